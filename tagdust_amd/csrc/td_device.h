// td_device.h -- device-side data layout shared by the HIP kernels (td_kernels.hip) and the C-ABI host
// layer (td_api.hip).  gfx950 only: wave64, one read per lane.
#pragma once
#include <stdint.h>

#define TD_WAVE 64
// live part of the logsum table: logsum() only indexes it when (max-min) < 15.7f, and
// (int)(15.699999f * 1000.0f) == 15699 (reference src/misc.c:72-78)
#define TD_LOGSUM_LIVE 15700

// per-column flags: bit k set <=> parameter k is live (!= -inf), so the term it gates can contribute
enum : uint32_t {
	TDF_MM = 1u << 0, TDF_MI = 1u << 1, TDF_MD = 1u << 2, TDF_II = 1u << 3, TDF_IM = 1u << 4,
	TDF_DD = 1u << 5, TDF_DM = 1u << 6, TDF_MSKIP = 1u << 7, TDF_ISKIP = 1u << 8,
	TDF_SM = 1u << 9, TDF_SI = 1u << 10,
};

// One HMM column: everything the DP reads for it, 96 bytes, read with wave-uniform (scalar) loads.
struct TdCol {
	float    t[9];     // transition[MM,MI,MD,II,IM,DD,DM,MSKIP,ISKIP]   (barcode_hmm.h:87-96)
	float    sM, sI;   // silent_to_M[f][g], silent_to_I[f][g]
	float    eM[5];    // m_emit
	float    eI[5];    // i_emit
	uint32_t flags;
	uint32_t pad[2];
};
static_assert(sizeof(TdCol) == 96, "TdCol layout");

struct TdSeg {
	int32_t n_hmm, n_col, col_off, hmm_off;
	float   skip;
	int32_t skip_live;   // skip != -inf
	int32_t type;        // 'B','R',...
	int32_t pad;
};

// per-HMM info for extraction (extract_reads, barcode_hmm.c:3205-3226)
//   bits 0..7 segment type char, 8..15 segment index, 16..30 hmm index in segment, 31 = is the all-N decoy of a B segment
struct TdModelHeader {
	int32_t S, H, C, avg_len;
	float   bg[5];
	float   r_stay;      // log(1 - 1/avg_len)    (barcode_hmm.c:4520)
	float   r_exit;      // log(1/avg_len)        (barcode_hmm.c:4523)
	int32_t required_finger_len;
	int32_t max_ncol;
	int32_t pad[3];
	TdSeg   seg[64];
};

// Per-wave-slot workspace layout (byte offsets from the slot base); every array is lane-interleaved
// [...][64 lanes] so that each wave-level access is one contiguous 64/256/512-byte run.
struct TdWsLayout {
	int64_t slot_bytes;
	int64_t codes;   // u8   [lmax+2][64]            unpacked base codes x_0..x_{lmax+1}
	int64_t sb;      // f32  [S+1][lmax+2][64]       silent_backward rows; row S = previous_silent
	int64_t sf;      // f32  [S+1][lmax+2][64]       row 0 = previous_silent, row j+1 = silent_forward of segment j
	int64_t bw;      // f32x2[C][lmax][64]           (M_backward, I_backward)[col][i-1]
	int64_t fwrow;   // f32x2[max_ncol][64]          forward row i-1 for segments too long for registers
	int64_t dp;      // f32  [lmax][H][64]           posterior label probabilities, rows i = 1..lmax
	int64_t path;    // u8   [lmax][H][64]
	int64_t acc;     // f32  [H][64]                 running label-DP row
	int64_t total;   // f32  [H][64]                 total_prob[h]
	int64_t dust;    // u8   [64][64]                DUST triplet counters
};

struct TdKernelArgs {
	const TdModelHeader* __restrict__ hdr;
	const TdCol*  __restrict__ cols;       // [C]
	const uint32_t* __restrict__ hinfo;    // [H]
	const int32_t*  __restrict__ pred_off; // [H+1]  CSR of allowed label predecessors u < v (A[u][v] == 1)
	const int32_t*  __restrict__ pred_idx;
	const float*  __restrict__ logsum;     // [TD_LOGSUM_LIVE] in HBM, staged into LDS per workgroup
	// batch (tile = 64 consecutive reads)
	const uint32_t* __restrict__ packed;   // [n_tiles][nw2 + nw1][64]  2-bit words then N-mask words
	const int32_t*  __restrict__ lens;     // [n_tiles*64]  (0 = padding lane)
	int32_t n_tiles, n_slots, lmax, nw2, nw1;
	int32_t mode;                          // TD_MODE_*
	float   threshold;
	int32_t minlen, dust;
	int32_t want_labels;
	int32_t win_start, win_len;             // -start / -end window for the DP phases; win_len = 0: whole reads
	// outputs
	float*   __restrict__ out_f;           // [n_tiles*64] each
	float*   __restrict__ out_b;
	float*   __restrict__ out_r;
	float*   __restrict__ out_bar;
	float*   __restrict__ out_q;
	int32_t* __restrict__ out_type;
	int32_t* __restrict__ out_barcode;
	int32_t* __restrict__ out_finger;
	uint32_t* __restrict__ out_keep;       // [n_tiles][nw1][64]  bit k of read = position k is kept (label is an R segment)
	int8_t*  __restrict__ out_labels;      // [n_tiles][lmax+1][64]
	unsigned long long* __restrict__ counters; // [TD_NUM_COUNTERS]
	// -ref artifact filter (td_artifact.inc); art_n == 0: off
	const uint8_t*  __restrict__ art_text;  // struct fasta ->string as read_fasta() leaves it
	const int32_t*  __restrict__ art_index; // [art_n + 1]
	const uint8_t*  __restrict__ art_left;  // [n_tiles*64] 1 = left-over read of its thread range (bpm_check_error path)
	int32_t art_n, art_fe;
	// workspace
	uint8_t* __restrict__ ws;
	TdWsLayout lay;
};

// ---------------------------------------------------------------------------------------------------------
// model-specialised kernel (td_spec_kernel.inc, compiled per model with hiprtc)
// ---------------------------------------------------------------------------------------------------------
struct TdSpecLayout {
	int64_t slot_bytes;
	int64_t codes;   // u8    [lmax+2][64]
	int64_t sb;      // f32   [S+1][lmax+2][64]
	int64_t sf;      // f32   [S+1][lmax+2][64]
	int64_t bw;      // per HMM h: [lmax][stored cols][64] x (M_backward, I_backward) = 8 B; float4 pairs + float2 tail
	int64_t dp;      // f32   [lmax][H][64]
	int64_t path;    // u32   [lmax][ceil(H/4)][64]   four path bytes per word
	int64_t total;   // f32   [H][64]
	int64_t rs;      // f32   [2*C][64]  restarted sweeps: (M, I) rows of every column at the position a regular sweep takes over
	int64_t bm;      // f32   [lmax+2][64]  running maximum of the first segment's label sums (kFirstN > 0)
	int64_t ba;      // u8    [lmax+2][64]  the label holding it
	int64_t acc;     // f32   [2][H][64]  label-DP rows (previous / current position) when too many labels for registers
	int64_t pmask;   // u32   [H][ceil(lmax/32)]  per wave, not per lane: bit i-1 of label h = the posterior row (h, i) was stored
	                 //       (some read of the tile has a non-zero posterior there); rows never stored are read as 0
};

struct TdSpecArgs {
	const float*  __restrict__ logsum;
	const uint32_t* __restrict__ packed;
	const int32_t*  __restrict__ lens;
	int32_t n_tiles, n_slots, lmax, nw2, nw1;
	int32_t mode;
	float   threshold;
	int32_t minlen, dust;
	int32_t win_start, win_len;             // -start / -end window for the DP phases; win_len = 0: whole reads
	int32_t prune_stride;                   // floats per table in `prune`
	float*   __restrict__ out_f;
	float*   __restrict__ out_b;
	float*   __restrict__ out_r;
	float*   __restrict__ out_bar;
	float*   __restrict__ out_q;
	int32_t* __restrict__ out_type;
	int32_t* __restrict__ out_barcode;
	int32_t* __restrict__ out_finger;
	uint32_t* __restrict__ out_keep;
	int8_t*  __restrict__ out_labels;
	unsigned long long* __restrict__ counters;
	// -ref artifact filter (td_artifact.inc); art_n == 0: off
	const uint8_t*  __restrict__ art_text;  // struct fasta ->string as read_fasta() leaves it
	const int32_t*  __restrict__ art_index; // [art_n + 1]
	const uint8_t*  __restrict__ art_left;  // [n_tiles*64] 1 = left-over read of its thread range (bpm_check_error path)
	int32_t art_n, art_fe;
	uint8_t* __restrict__ ws;
	TdSpecLayout lay;
	// position pruning (td_spec_kernel.inc): four host tables of prune_stride floats each -- fb[i], bwb[m], wa[i], wb[i]
	const float* __restrict__ prune;
	int32_t* __restrict__ tile_next;   // tiles handed out so far beyond the first n_slots (dynamic tile assignment): zero before the launch
	// Length classes: a batch whose longest reads are far longer than the rest (one 1000-base read among 150-base ones) keeps the
	// workspace geometry of the many.  Wave slots [0, n_big) are laid out for the batch's longest read (lay_big, lmax_big) and
	// sit in front of the others in `ws`; slots take their first tile by slot number, longest tiles first, so the n_long <= n_big
	// tiles longer than `lmax` (the geometry of the other slots) all start in a big slot.  n_big = 0: one geometry (lay, lmax).
	int32_t n_big, lmax_big;
	int32_t out_lmax;                  // stride of out_labels (the batch's longest read), whatever geometry a slot has
	TdSpecLayout lay_big;
	// the runs of every read's labels[0..len] -- (run length << 8) | label, at most rle_cap per read, unused entries 0 -- for the
	// compact egress (td_api.hip): [n_tiles][rle_cap][64]; *rle_overflow is set when a read has more runs.  nullptr: not wanted
	uint32_t* __restrict__ out_runs;
	int32_t*  __restrict__ rle_overflow;
	int32_t rle_cap;
};
