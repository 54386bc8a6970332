// td_stage.h -- device-side ingest and egress around the decode kernels (td_stage.hip): what the reference does on the
// host before and after run_pHMM()'s thread fan-out -- base coding (init_nuc_code, src/nuc_code.c:46-74), the read_info
// arrays (src/io.h:76-91) and the in-place rewrite of ri->seq (make_extracted_read, src/barcode_hmm.c:3325-3356) --
// laid out for the one-read-per-lane kernels.  Library-internal.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// One batch as it moves through the device.  "Device order" k = position of a read after the stable length sort
// (tile = k / 64, lane = k % 64); read_at[k] is the read's index in the caller's order (nullptr: identity, all reads
// have one length).
struct TdStageBatch {
	// inputs in the caller's order
	const uint8_t* raw;       // [offs[n]] base codes 0..4 or ASCII sequence text
	const int64_t* offs;      // [n + 1]
	int64_t n_reads;
	int32_t is_ascii;
	int32_t n_tiles, lmax, nw2, nw1;
	const int32_t* read_at;   // [n] or nullptr
	// -ref artifact filter: reads of each thread range are taken in fours, the remainder goes to another routine
	// (match_to_reference, src/barcode_hmm.c:2495-2575; ranges as in run_pHMM :1911-1922).  art_threads = 0: off
	int32_t art_threads;
	int64_t art_first, art_total;   // this batch is reads [art_first, art_first + n_reads) of a batch of art_total (0: it is the whole batch)
	// decode-kernel side (device order)
	uint32_t* packed;         // [n_tiles][nw2 + nw1][64]
	int32_t*  lens;           // [n_tiles * 64]
	uint8_t*  art_left;       // [n_tiles * 64]
	const uint8_t* out_soa;   // f, b, r, bar, q, type, barcode, finger: 8 arrays of n_tiles*64 x 4 B at soa_stride apart
	int64_t   soa_stride;
	const uint32_t* keep;     // [n_tiles][nw1][64]
	const int8_t*  labels;    // [n_tiles][lmax + 1][64]
	const uint32_t* runs;     // [n_tiles][rle_cap][64] label runs as the specialised kernel leaves them, or nullptr (then rle_out is scanned from labels)
	// results in the caller's order (any may be nullptr)
	uint8_t* res;             // [n] td_read_result (32 B)
	uint8_t* seq_out;         // [offs[n]]
	int8_t*  labels_out;      // [offs[n] + n]
	// ... or, for the two big ones, what the host can rebuild them from (a fifth of the bytes over PCIe):
	uint32_t* keep_out;       // [n][nw1] the keep bits of read i (bit p: base p stays, else it becomes the spacer byte 65)
	uint32_t* rle_out;        // [n][rle_cap] the runs of ri->labels[0..len]: (run length << 8) | label, unused entries 0
	int32_t   rle_cap;
	int32_t*  rle_overflow;   // set to 1 when a read has more than rle_cap runs (the host then asks for labels_out)
	const int32_t* runs_overflow;   // the decode kernel's own flag of that kind (with runs), folded into rle_overflow: one word to fetch
};

// bytes of scratch td_stage_sort needs for n reads
size_t td_stage_sort_temp_bytes(int64_t n_reads, int lmax);
// read_at <- stable sort of the read indices by length; keys / keys_alt / vals_alt: [n] scratch, temp: td_stage_sort_temp_bytes
hipError_t td_stage_sort(const int64_t* offs, int64_t n_reads, int lmax, int32_t* read_at, uint32_t* keys, uint32_t* keys_alt,
                         int32_t* vals_alt, void* temp, size_t temp_bytes, hipStream_t stream);
// raw bases -> 2-bit words + N mask, lane-interleaved per tile; lens
hipError_t td_stage_pack(const TdStageBatch& b, hipStream_t stream);
// art_left[k] = read k is a left-over read of its thread range (art_threads ranges over the caller's order)
hipError_t td_stage_art_left(const TdStageBatch& b, hipStream_t stream);
// SoA / lane-interleaved kernel outputs -> per-read records, rewritten sequences and labels in the caller's order
hipError_t td_stage_finish(const TdStageBatch& b, hipStream_t stream);
// memory-side probe of a candidate workspace allocation (milliseconds, best of three passes); see td_stage.hip
hipError_t td_ws_probe(uint8_t* ws, int64_t slot_bytes, int n_slots, hipStream_t stream, float* ms);
