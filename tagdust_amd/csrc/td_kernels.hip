// td_kernels.hip -- TagDust2 per-read HMM decoding on gfx950 (MI355X), hand-written HIP: the GENERIC ahead-of-time
// kernel, which reads the model from HBM (any model the ABI accepts, no per-model compile).  The default path is the
// model-specialised kernel in td_spec_kernel.inc; this one serves td_set_option("specialize", 0) and the candidate
// models of an architecture comparison.
//
// Mapping: ONE READ PER LANE.  A wave owns a tile of 64 reads and walks the read-architecture HMM with
// wave-uniform control flow (segment j, HMM f, position i, column g are scalar loop counters), so every
// vector instruction does 64 reads' worth of useful work and no cross-lane operation exists anywhere.
// That is what bit-exactness demands: the reference's logsum() is a table-quantised operator, every
// "(+)" chain below must be folded in the reference's order, and a shuffle/tree reduction over lanes
// would re-associate it (SURVEY.md 3.4, 7 "Hard parts").
//
//   - logsum table (15 700 live floats, 62.8 KB) staged in LDS once per workgroup; one ds_read_b32 per logsum
//   - model parameters are wave-uniform -> scalar loads (TdCol), terms gated by -inf parameters are
//     skipped with scalar branches (logsum(x, -inf) == x exactly)
//   - per-HMM DP rows (row i+1 going backward, row i-1 going forward) live in registers (templated on
//     the segment's column count, <= 16; longer segments fall back to rows kept in the workspace)
//   - the backward rows the forward pass needs (M_backward, I_backward for every column and position)
//     stream to a per-wave HBM workspace as lane-interleaved float2 (512 B per wave store) and stream
//     back once: this spill is the kernel's real HBM traffic
//
// Reference: backward() src/barcode_hmm.c:3439-3640, forward_max_posterior_decoding() :4128-4525,
// Q value do_label_thread :2320-2338, extract_reads()/make_extracted_read() :3172-3356,
// dust_sequences() :2407-2467, logsum() src/misc.c:72-78.
#include <hip/hip_runtime.h>
#include <math.h>
#include "td_device.h"
#include "td_artifact.inc"

#define TD_BLOCK 256                 // 4 waves share one LDS copy of the logsum table
#define TD_WAVES_PER_BLOCK (TD_BLOCK / TD_WAVE)
#define TD_MAX_REG_NCOL 16

#define NEG_INF (-__builtin_inff())

// run modes / outcomes (mirrors include/tagdust_hip.h)
#define MODE_GET_LABEL 1
#define MODE_GET_PROB 4
#define MODE_ARCH_COMP 5
#define OUT_SUCCESS 0
#define OUT_ARCH_MISMATCH 1
#define OUT_TOO_SHORT 2
#define OUT_BAR_FINGER_NOT_FOUND 3
#define OUT_MATCHES_ARTIFACTS 5
#define OUT_LOW_COMPLEXITY 6
#define N_OUTCOME_SLOTS 8

// ---------------------------------------------------------------------------------------------------------
// logsum, src/misc.c:72-78:  (min == -inf || max-min >= 15.7f) ? max : max + T[(int)((max-min)*1000.0f)]
// 8 VALU + 1 ds_read_b32, no select: every "return max" case lands on a table slot that holds 0.
// ---------------------------------------------------------------------------------------------------------
__shared__ float g_T[TD_LOGSUM_LIVE + 4]; // slot [TD_LOGSUM_LIVE] holds 0: "return max" without a select

struct LdsTable {};   // tag: the table is the file-scope LDS array, never a generic pointer (a generic pointer
                      // turns every lookup into a flat_load that waits on vmcnt AND lgkmcnt)

__device__ __forceinline__ float lsum(LdsTable, float a, float b)
{
	// d >= 15.7f, d = +inf (one operand -inf) and d = NaN (both -inf) clamp to 15.7f -> index 15700 -> + 0.0f
	const float mx = fmaxf(a, b);
	const float dc = fminf(mx - fminf(a, b), 15.7f);
	return mx + g_T[(int)(dc * 1000.0f)];
}

// Model tables are read through the constant address space: with a wave-uniform index the compiler then
// emits scalar loads (s_load_dword*) instead of per-lane global loads.
#define TD_CONST __attribute__((address_space(4)))
typedef const TdCol TD_CONST* ccol_ptr;
typedef const TdModelHeader TD_CONST* chdr_ptr;
typedef const float TD_CONST* cfloat_ptr;
typedef const int32_t TD_CONST* cint_ptr;
typedef const uint32_t TD_CONST* cuint_ptr;
template <typename T> __device__ __forceinline__ const T TD_CONST* as_const(const T* p)
{
	return (const T TD_CONST*)(uintptr_t)p;
}

__device__ __forceinline__ TdSeg load_seg(chdr_ptr hd, int j)
{
	TdSeg s;
	s.n_hmm = hd->seg[j].n_hmm; s.n_col = hd->seg[j].n_col; s.col_off = hd->seg[j].col_off; s.hmm_off = hd->seg[j].hmm_off;
	s.skip = hd->seg[j].skip; s.skip_live = hd->seg[j].skip_live; s.type = hd->seg[j].type; s.pad = 0;
	return s;
}

// emission lookup with a per-lane base code and wave-uniform table (5 scalars)
__device__ __forceinline__ float emit5(cfloat_ptr e, int c)
{
	// the five values are pinned in SGPRs first: otherwise the compiler folds "select of loads" into a per-lane
	// indexed load from the table, i.e. a vector memory access per emission look-up
	float e0 = e[0], e1 = e[1], e2 = e[2], e3 = e[3], e4 = e[4];
	asm volatile("" : "+s"(e0), "+s"(e1), "+s"(e2), "+s"(e3), "+s"(e4));
	float r = e4;
	r = (c == 3) ? e3 : r;
	r = (c == 2) ? e2 : r;
	r = (c == 1) ? e1 : r;
	r = (c == 0) ? e0 : r;
	return r;
}

struct WaveCtx {
	uint8_t* slot;         // this wave's workspace slot
	int lane;
	int len;               // this lane's read length (0 = idle lane)
	int lmax;              // batch-wide row stride (positions)
	int tmax;              // max len within this tile (wave-uniform)
};

__device__ __forceinline__ float* ws_f32(const WaveCtx& w, int64_t off) { return (float*)(w.slot + off); }

// ---------------------------------------------------------------------------------------------------------
// backward, one HMM (segment j, hmm f), rows in registers.  barcode_hmm.c:3505-3607
//   P  = silent_backward of segment j+1 (or previous_silent), Cs = silent_backward of segment j
// ---------------------------------------------------------------------------------------------------------
template <int NCOL>
__device__ __forceinline__ void bwd_hmm_reg(const WaveCtx& w, ccol_ptr cp,
                                            const uint8_t* __restrict__ codes, const float* __restrict__ P,
                                            float* __restrict__ Cs, float2* __restrict__ bw,
                                            bool first_f, bool skip_live, float skipj)
{
	constexpr int K = NCOL - 1;
	const LdsTable T{};
	float Mn[NCOL], In[NCOL];
#pragma unroll
	for (int g = 0; g < NCOL; g++) { Mn[g] = NEG_INF; In[g] = NEG_INF; }
	int c = 0; // x_{i+1}; the reference reads the 0 terminator for i == len (only ever added to -inf terms)

	for (int i = w.tmax; i >= 1; --i) {
		if (i <= w.len) {
			const int xi = codes[i * TD_WAVE + w.lane];
			const float Pn = P[(i + 1) * TD_WAVE + w.lane];
			float cs = first_f ? NEG_INF : Cs[i * TD_WAVE + w.lane];
			float Mc[NCOL], Ic[NCOL];
			float Dp;
			{   // last column, :3518-3543
				const TdCol TD_CONST& q = cp[K];
				const uint32_t fl = q.flags;
				float M = Pn + q.t[7];
				float I = Pn + q.t[8];
				if (fl & TDF_IM) I = lsum(T, I, (Mn[K] + q.t[4]) + emit5(q.eM, c));
				if (fl & TDF_II) I = lsum(T, I, (In[K] + q.t[3]) + emit5(q.eI, c));
				if (fl & TDF_SM) cs = lsum(T, cs, (M + q.sM) + emit5(q.eM, xi));
				if (fl & TDF_SI) cs = lsum(T, cs, (I + q.sI) + emit5(q.eI, xi));
				Mc[K] = M; Ic[K] = I; Dp = NEG_INF;
			}
#pragma unroll
			for (int g = K - 1; g >= 0; --g) { // :3544-3586
				const TdCol TD_CONST& q = cp[g];
				const TdCol TD_CONST& qp = cp[g + 1];
				const uint32_t fl = q.flags;
				const float epc = emit5(qp.eM, c);
				const float eic = emit5(q.eI, c);
				float M = (Mn[g + 1] + epc) + q.t[0];
				if (fl & TDF_MSKIP) M = lsum(T, M, Pn + q.t[7]);
				if (fl & TDF_MI) M = lsum(T, M, (In[g] + eic) + q.t[1]);
				if (fl & TDF_MD) M = lsum(T, M, Dp + q.t[2]);
				float I = (In[g] + q.t[3]) + eic;
				if (fl & TDF_ISKIP) I = lsum(T, I, Pn + q.t[8]);
				if (fl & TDF_IM) I = lsum(T, I, (Mn[g + 1] + q.t[4]) + epc);
				float D = Dp + q.t[5];
				if (fl & TDF_DM) D = lsum(T, D, (Mc[g + 1] + emit5(qp.eM, xi)) + q.t[6]);
				if (fl & TDF_SM) cs = lsum(T, cs, (M + q.sM) + emit5(q.eM, xi));
				if (fl & TDF_SI) cs = lsum(T, cs, (I + q.sI) + emit5(q.eI, xi));
				Mc[g] = M; Ic[g] = I; Dp = D;
			}
			if (skip_live) cs = lsum(T, cs, P[i * TD_WAVE + w.lane] + skipj); // :3600 (once per HMM: reference quirk)
			Cs[i * TD_WAVE + w.lane] = cs;
#pragma unroll
			for (int g = 0; g < NCOL; g++) {
				bw[((int64_t)g * w.lmax + (i - 1)) * TD_WAVE + w.lane] = make_float2(Mc[g], Ic[g]);
				Mn[g] = Mc[g]; In[g] = Ic[g];
			}
			c = xi;
		}
	}
}

// Same recurrence for segments longer than TD_MAX_REG_NCOL columns: row i+1 is re-read from the
// spilled backward rows (they are exactly what the previous iteration stored).
__device__ __forceinline__ void bwd_hmm_mem(const WaveCtx& w, int ncol, ccol_ptr cp,
                                         const uint8_t* __restrict__ codes, const float* __restrict__ P,
                                         float* __restrict__ Cs, float2* __restrict__ bw,
                                         bool first_f, bool skip_live, float skipj)
{
	const int K = ncol - 1;
	const LdsTable T{};
	int c = 0;
	for (int i = w.tmax; i >= 1; --i) {
		if (i <= w.len) {
			const bool top = (i == w.len);
			const int xi = codes[i * TD_WAVE + w.lane];
			const float Pn = P[(i + 1) * TD_WAVE + w.lane];
			float cs = first_f ? NEG_INF : Cs[i * TD_WAVE + w.lane];
			float Mcp, Dp; // M, D of column g+1 at row i
			float2 nxp;    // (M,I) of column g+1 at row i+1
			{
				const TdCol TD_CONST& q = cp[K];
				const uint32_t fl = q.flags;
				nxp = top ? make_float2(NEG_INF, NEG_INF) : bw[((int64_t)K * w.lmax + i) * TD_WAVE + w.lane];
				float M = Pn + q.t[7];
				float I = Pn + q.t[8];
				if (fl & TDF_IM) I = lsum(T, I, (nxp.x + q.t[4]) + emit5(q.eM, c));
				if (fl & TDF_II) I = lsum(T, I, (nxp.y + q.t[3]) + emit5(q.eI, c));
				if (fl & TDF_SM) cs = lsum(T, cs, (M + q.sM) + emit5(q.eM, xi));
				if (fl & TDF_SI) cs = lsum(T, cs, (I + q.sI) + emit5(q.eI, xi));
				bw[((int64_t)K * w.lmax + (i - 1)) * TD_WAVE + w.lane] = make_float2(M, I);
				Mcp = M; Dp = NEG_INF;
			}
			for (int g = K - 1; g >= 0; --g) {
				const TdCol TD_CONST& q = cp[g];
				const TdCol TD_CONST& qp = cp[g + 1];
				const uint32_t fl = q.flags;
				const float2 nx = top ? make_float2(NEG_INF, NEG_INF) : bw[((int64_t)g * w.lmax + i) * TD_WAVE + w.lane];
				const float epc = emit5(qp.eM, c);
				const float eic = emit5(q.eI, c);
				float M = (nxp.x + epc) + q.t[0];
				if (fl & TDF_MSKIP) M = lsum(T, M, Pn + q.t[7]);
				if (fl & TDF_MI) M = lsum(T, M, (nx.y + eic) + q.t[1]);
				if (fl & TDF_MD) M = lsum(T, M, Dp + q.t[2]);
				float I = (nx.y + q.t[3]) + eic;
				if (fl & TDF_ISKIP) I = lsum(T, I, Pn + q.t[8]);
				if (fl & TDF_IM) I = lsum(T, I, (nxp.x + q.t[4]) + epc);
				float D = Dp + q.t[5];
				if (fl & TDF_DM) D = lsum(T, D, (Mcp + emit5(qp.eM, xi)) + q.t[6]);
				if (fl & TDF_SM) cs = lsum(T, cs, (M + q.sM) + emit5(q.eM, xi));
				if (fl & TDF_SI) cs = lsum(T, cs, (I + q.sI) + emit5(q.eI, xi));
				bw[((int64_t)g * w.lmax + (i - 1)) * TD_WAVE + w.lane] = make_float2(M, I);
				Mcp = M; Dp = D; nxp = nx;
			}
			if (skip_live) cs = lsum(T, cs, P[i * TD_WAVE + w.lane] + skipj);
			Cs[i * TD_WAVE + w.lane] = cs;
			c = xi;
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
// forward + posterior accumulation, one HMM, rows in registers.  barcode_hmm.c:4213-4344
//   P = silent_forward of segment j-1 (or previous_silent), Cs = silent_forward of segment j
//   dp row i gets the label posterior of HMM h as a probability: scaledprob2prob() (:4431-4440)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float post_prob(float lp)
{
	// scaledprob2prob, src/misc.c:98-105: exp() in double on the float argument, result narrowed to float
	return (lp == NEG_INF) ? 0.0f : (float)exp((double)lp);
}

template <int NCOL>
__device__ __forceinline__ float fwd_hmm_reg(const WaveCtx& w, ccol_ptr cp,
                                             const uint8_t* __restrict__ codes, const float* __restrict__ P,
                                             float* __restrict__ Cs, const float2* __restrict__ bw,
                                             float* __restrict__ dp_h, int H, float b,
                                             bool first_f, bool skip_live, float skipj)
{
	constexpr int K = NCOL - 1;
	const LdsTable T{};
	float Mp[NCOL], Ip[NCOL];
#pragma unroll
	for (int g = 0; g < NCOL; g++) { Mp[g] = NEG_INF; Ip[g] = NEG_INF; }
	float tot = NEG_INF;

	for (int i = 1; i <= w.tmax; ++i) {
		if (i <= w.len) {
			const int c = codes[i * TD_WAVE + w.lane];
			const float Pm = P[(i - 1) * TD_WAVE + w.lane];
			float cs = first_f ? NEG_INF : Cs[i * TD_WAVE + w.lane];
			float Mc[NCOL], Ic[NCOL];
			float Dc;
			float acc = NEG_INF;
			{   // column 0, :4220-4268
				const TdCol TD_CONST& q = cp[0];
				const uint32_t fl = q.flags;
				const float2 B = bw[((int64_t)0 * w.lmax + (i - 1)) * TD_WAVE + w.lane];
				const float em = emit5(q.eM, c);
				const float ei = emit5(q.eI, c);
				const float M = (Pm + q.sM) + em;
				const float pm = (M + B.x) - b;
				tot = lsum(T, tot, pm);
				acc = lsum(T, acc, pm);
				float I = Pm + q.sI;
				if (fl & TDF_II) I = lsum(T, I, Ip[0] + q.t[3]);
				if (fl & TDF_MI) I = lsum(T, I, Mp[0] + q.t[1]);
				I = I + ei;
				if (fl & TDF_SI) tot = lsum(T, tot, (((Pm + q.sI) + ei) + B.y) - b);
				acc = lsum(T, acc, (I + B.y) - b);
				Dc = NEG_INF;
				if (fl & TDF_MSKIP) cs = lsum(T, cs, M + q.t[7]);
				if (fl & TDF_ISKIP) cs = lsum(T, cs, I + q.t[8]);
				Mc[0] = M; Ic[0] = I;
			}
#pragma unroll
			for (int g = 1; g <= K; ++g) { // :4271-4334
				const TdCol TD_CONST& q = cp[g];
				const TdCol TD_CONST& qp = cp[g - 1];
				const uint32_t fl = q.flags, flp = qp.flags;
				const float2 B = bw[((int64_t)g * w.lmax + (i - 1)) * TD_WAVE + w.lane];
				float M = Pm + q.sM;
				if (flp & TDF_MM) M = lsum(T, M, Mp[g - 1] + qp.t[0]);
				if (flp & TDF_IM) M = lsum(T, M, Ip[g - 1] + qp.t[4]);
				if (flp & TDF_DM) M = lsum(T, M, Dc + qp.t[6]);
				M = M + emit5(q.eM, c);
				acc = lsum(T, acc, (M + B.x) - b);
				float I = Pm + q.sI;
				if (fl & TDF_II) I = lsum(T, I, Ip[g] + q.t[3]);
				if (fl & TDF_MI) I = lsum(T, I, Mp[g] + q.t[1]);
				I = I + emit5(q.eI, c);
				acc = lsum(T, acc, (I + B.y) - b);
				float D = Mc[g - 1] + qp.t[2];
				if (flp & TDF_DD) D = lsum(T, D, Dc + qp.t[5]);
				if (fl & TDF_MSKIP) cs = lsum(T, cs, M + q.t[7]);
				if (fl & TDF_ISKIP) cs = lsum(T, cs, I + q.t[8]);
				Mc[g] = M; Ic[g] = I; Dc = D;
			}
			if (skip_live) cs = lsum(T, cs, P[i * TD_WAVE + w.lane] + skipj); // :4341
			Cs[i * TD_WAVE + w.lane] = cs;
			dp_h[(int64_t)(i - 1) * H * TD_WAVE + w.lane] = post_prob(acc);
#pragma unroll
			for (int g = 0; g < NCOL; g++) { Mp[g] = Mc[g]; Ip[g] = Ic[g]; }
		}
	}
	return tot;
}

// long segments: row i-1 kept in the workspace (fwrow[g] = (M_forward, I_forward)[g][i-1])
__device__ __forceinline__ float fwd_hmm_mem(const WaveCtx& w, int ncol, ccol_ptr cp,
                                          const uint8_t* __restrict__ codes, const float* __restrict__ P,
                                          float* __restrict__ Cs, const float2* __restrict__ bw,
                                          float2* __restrict__ fwrow, float* __restrict__ dp_h, int H, float b,
                                          bool first_f, bool skip_live, float skipj)
{
	const int K = ncol - 1;
	const LdsTable T{};
	for (int g = 0; g < ncol; g++) fwrow[g * TD_WAVE + w.lane] = make_float2(NEG_INF, NEG_INF);
	float tot = NEG_INF;
	for (int i = 1; i <= w.tmax; ++i) {
		if (i <= w.len) {
			const int c = codes[i * TD_WAVE + w.lane];
			const float Pm = P[(i - 1) * TD_WAVE + w.lane];
			float cs = first_f ? NEG_INF : Cs[i * TD_WAVE + w.lane];
			float acc = NEG_INF;
			float Mcp, Dc;      // M, D of column g-1 at row i
			float2 pvp;         // (M,I) of column g-1 at row i-1
			{
				const TdCol TD_CONST& q = cp[0];
				const uint32_t fl = q.flags;
				const float2 B = bw[((int64_t)0 * w.lmax + (i - 1)) * TD_WAVE + w.lane];
				pvp = fwrow[0 * TD_WAVE + w.lane];
				const float em = emit5(q.eM, c);
				const float ei = emit5(q.eI, c);
				const float M = (Pm + q.sM) + em;
				const float pm = (M + B.x) - b;
				tot = lsum(T, tot, pm);
				acc = lsum(T, acc, pm);
				float I = Pm + q.sI;
				if (fl & TDF_II) I = lsum(T, I, pvp.y + q.t[3]);
				if (fl & TDF_MI) I = lsum(T, I, pvp.x + q.t[1]);
				I = I + ei;
				if (fl & TDF_SI) tot = lsum(T, tot, (((Pm + q.sI) + ei) + B.y) - b);
				acc = lsum(T, acc, (I + B.y) - b);
				Dc = NEG_INF;
				if (fl & TDF_MSKIP) cs = lsum(T, cs, M + q.t[7]);
				if (fl & TDF_ISKIP) cs = lsum(T, cs, I + q.t[8]);
				fwrow[0 * TD_WAVE + w.lane] = make_float2(M, I);
				Mcp = M;
			}
			for (int g = 1; g <= K; ++g) {
				const TdCol TD_CONST& q = cp[g];
				const TdCol TD_CONST& qp = cp[g - 1];
				const uint32_t fl = q.flags, flp = qp.flags;
				const float2 B = bw[((int64_t)g * w.lmax + (i - 1)) * TD_WAVE + w.lane];
				const float2 pv = fwrow[g * TD_WAVE + w.lane];
				float M = Pm + q.sM;
				if (flp & TDF_MM) M = lsum(T, M, pvp.x + qp.t[0]);
				if (flp & TDF_IM) M = lsum(T, M, pvp.y + qp.t[4]);
				if (flp & TDF_DM) M = lsum(T, M, Dc + qp.t[6]);
				M = M + emit5(q.eM, c);
				acc = lsum(T, acc, (M + B.x) - b);
				float I = Pm + q.sI;
				if (fl & TDF_II) I = lsum(T, I, pv.y + q.t[3]);
				if (fl & TDF_MI) I = lsum(T, I, pv.x + q.t[1]);
				I = I + emit5(q.eI, c);
				acc = lsum(T, acc, (I + B.y) - b);
				float D = Mcp + qp.t[2];
				if (flp & TDF_DD) D = lsum(T, D, Dc + qp.t[5]);
				if (fl & TDF_MSKIP) cs = lsum(T, cs, M + q.t[7]);
				if (fl & TDF_ISKIP) cs = lsum(T, cs, I + q.t[8]);
				fwrow[g * TD_WAVE + w.lane] = make_float2(M, I);
				Mcp = M; Dc = D; pvp = pv;
			}
			if (skip_live) cs = lsum(T, cs, P[i * TD_WAVE + w.lane] + skipj);
			Cs[i * TD_WAVE + w.lane] = cs;
			dp_h[(int64_t)(i - 1) * H * TD_WAVE + w.lane] = post_prob(acc);
		}
	}
	return tot;
}

#define TD_DISPATCH_NCOL(N, CALL_REG, CALL_MEM)                                                    \
	switch (N) {                                                                                   \
	case 1: { constexpr int NC = 1; CALL_REG; } break;   case 2: { constexpr int NC = 2; CALL_REG; } break;   \
	case 3: { constexpr int NC = 3; CALL_REG; } break;   case 4: { constexpr int NC = 4; CALL_REG; } break;   \
	case 5: { constexpr int NC = 5; CALL_REG; } break;   case 6: { constexpr int NC = 6; CALL_REG; } break;   \
	case 7: { constexpr int NC = 7; CALL_REG; } break;   case 8: { constexpr int NC = 8; CALL_REG; } break;   \
	case 9: { constexpr int NC = 9; CALL_REG; } break;   case 10: { constexpr int NC = 10; CALL_REG; } break; \
	case 11: { constexpr int NC = 11; CALL_REG; } break; case 12: { constexpr int NC = 12; CALL_REG; } break; \
	case 13: { constexpr int NC = 13; CALL_REG; } break; case 14: { constexpr int NC = 14; CALL_REG; } break; \
	case 15: { constexpr int NC = 15; CALL_REG; } break; case 16: { constexpr int NC = 16; CALL_REG; } break; \
	default: { CALL_MEM; } break;                                                                  \
	}

// one segment, all its HMMs (templated on the column count so the f loop stays a scalar loop)
template <int NCOL>
__device__ __forceinline__ void bwd_segment_reg(const WaveCtx& w, const TdSeg sg, ccol_ptr cols,
                                             const uint8_t* __restrict__ codes, const float* __restrict__ P,
                                             float* __restrict__ Cs, float2* __restrict__ bwbase)
{
	for (int f = 0; f < sg.n_hmm; f++) {
		const int col0 = sg.col_off + f * NCOL;
		bwd_hmm_reg<NCOL>(w, cols + col0, codes, P, Cs, bwbase + (int64_t)col0 * w.lmax * TD_WAVE,
		                  f == 0, sg.skip_live != 0, sg.skip);
	}
}

template <int NCOL>
__device__ __forceinline__ void fwd_segment_reg(const WaveCtx& w, const TdSeg sg, ccol_ptr cols,
                                             const uint8_t* __restrict__ codes, const float* __restrict__ P,
                                             float* __restrict__ Cs, const float2* __restrict__ bwbase,
                                             float* __restrict__ dp, float* __restrict__ total, int H, float b)
{
	for (int f = 0; f < sg.n_hmm; f++) {
		const int col0 = sg.col_off + f * NCOL;
		const int h = sg.hmm_off + f;
		const float tot = fwd_hmm_reg<NCOL>(w, cols + col0, codes, P, Cs, bwbase + (int64_t)col0 * w.lmax * TD_WAVE,
		                                    dp + (int64_t)h * TD_WAVE, H, b, f == 0, sg.skip_live != 0, sg.skip);
		total[h * TD_WAVE + w.lane] = tot;
	}
}

// ---------------------------------------------------------------------------------------------------------
// the kernel: a persistent wave walks tiles slot, slot + n_slots, ...
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void decode_body(const TdKernelArgs& ka, const int block_x)
{
	const LdsTable T{};
	for (int k = threadIdx.x; k < TD_LOGSUM_LIVE + 4; k += TD_BLOCK) g_T[k] = (k < TD_LOGSUM_LIVE) ? ka.logsum[k] : 0.0f;
	__syncthreads();

	const int lane = threadIdx.x & (TD_WAVE - 1);
	const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int slot = block_x * TD_WAVES_PER_BLOCK + wave_in_block;
	if (slot >= ka.n_slots) return;

	const TdModelHeader TD_CONST& hd = *as_const(ka.hdr);
	const ccol_ptr cols_c = as_const(ka.cols);
	const cint_ptr pred_off = as_const(ka.pred_off);
	const cint_ptr pred_idx = as_const(ka.pred_idx);
	const int S = hd.S, H = hd.H;
	const int lmax = ka.lmax;
	const int rowlen = (lmax + 2) * TD_WAVE; // floats per silent row

	WaveCtx w;
	w.slot = ka.ws + (int64_t)slot * ka.lay.slot_bytes;
	w.lane = lane;
	w.lmax = lmax;

	uint8_t* const codesF = w.slot + ka.lay.codes;   // base codes of the whole reads
	float* SB = ws_f32(w, ka.lay.sb);
	float* SF = ws_f32(w, ka.lay.sf);
	float2* BW = (float2*)(w.slot + ka.lay.bw);
	float2* FWROW = (float2*)(w.slot + ka.lay.fwrow);
	float* DP = ws_f32(w, ka.lay.dp);
	uint8_t* PATH = w.slot + ka.lay.path;
	float* ACC = ws_f32(w, ka.lay.acc);
	float* TOTAL = ws_f32(w, ka.lay.total);
	uint8_t* DUST = w.slot + ka.lay.dust;

	for (int tile = slot; tile < ka.n_tiles; tile += ka.n_slots) {
		const int64_t rid = (int64_t)tile * TD_WAVE + lane;
		// -start / -end (td_set_window): `len`, `tmax`, `codes` describe the window the DP phases see (do_label_thread decodes
		// seq + matchstart for matchend - matchstart bases, barcode_hmm.c:2290-2296); `lenF`, `tmaxF`, `codesF` the whole
		// read, which the rewrite, the artifact filter and DUST walk (make_extracted_read :3336)
		const int lenF = ka.lens[rid];
		const int woff = ka.win_len > 0 ? ka.win_start : 0;
		int len = lenF;
		if (ka.win_len > 0) {
			const int e = lenF < woff + ka.win_len ? lenF : woff + ka.win_len;
			len = e > woff ? e - woff : 0;
		}
		w.len = len;
		int tmax = len;
#pragma unroll
		for (int o = 32; o >= 1; o >>= 1) tmax = max(tmax, __shfl_xor(tmax, o));
		tmax = __builtin_amdgcn_readfirstlane(tmax);
		w.tmax = tmax;
		int tmaxF = lenF;
#pragma unroll
		for (int o = 32; o >= 1; o >>= 1) tmaxF = max(tmaxF, __shfl_xor(tmaxF, o));
		tmaxF = __builtin_amdgcn_readfirstlane(tmaxF);
		uint8_t* const codes = codesF + woff * TD_WAVE;

		// ---- phase 0: unpack 2-bit codes + N mask into x_0..x_{tmax+1} (x_0 and x_{>len} = 0) ----
		{
			const uint32_t* pk = ka.packed + (int64_t)tile * (ka.nw2 + ka.nw1) * TD_WAVE;
			codesF[lane] = 0;
			uint32_t w2 = 0, w1 = 0;
			for (int i = 1; i <= tmaxF + 1; i++) {
				const int k = i - 1;
				if ((k & 15) == 0) w2 = (k >> 4) < ka.nw2 ? pk[(k >> 4) * TD_WAVE + lane] : 0u;
				if ((k & 31) == 0) w1 = (k >> 5) < ka.nw1 ? pk[(ka.nw2 + (k >> 5)) * TD_WAVE + lane] : 0u;
				uint32_t code = (w2 >> (2 * (k & 15))) & 3u;
				if ((w1 >> (k & 31)) & 1u) code = 4u;
				if (i > lenF) code = 0u;
				codesF[i * TD_WAVE + lane] = (uint8_t)code;
			}
		}

		// ---- phase 1: backward, barcode_hmm.c:3439-3640 ----
		float b_score;
		{
			// init :3483-3492.  Row S = previous_silent: -inf except [len+1] = log(1) = 0.
			// Only entries that are read before being produced need initialising.
			float* prev = SB + (int64_t)S * rowlen;
			for (int i = 1; i <= tmax + 1; i++) prev[i * TD_WAVE + lane] = (i == len + 1) ? 0.0f : NEG_INF;
			float run = 0.0f;
			for (int j = S - 1; j >= 0; j--) {
				run = run + hd.seg[j].skip; // SB_{S-1}[L+1] = 0 + skip_{S-1}; SB_j[L+1] = SB_{j+1}[L+1] + skip_j
				SB[(int64_t)j * rowlen + (len + 1) * TD_WAVE + lane] = run;
			}
			for (int j = S - 1; j >= 0; j--) {
				const TdSeg sg = load_seg(&hd, j);
				const float* P = SB + (int64_t)(j + 1) * rowlen;
				float* Cs = SB + (int64_t)j * rowlen;
				TD_DISPATCH_NCOL(sg.n_col,
					bwd_segment_reg<NC>(w, sg, cols_c, codes, P, Cs, BW),
					for (int f = 0; f < sg.n_hmm; f++) {
						const int col0 = sg.col_off + f * sg.n_col;
						bwd_hmm_mem(w, sg.n_col, cols_c + col0, codes, P, Cs, BW + (int64_t)col0 * lmax * TD_WAVE,
						            f == 0, sg.skip_live != 0, sg.skip);
					})
			}
			b_score = (len >= 1) ? SB[0 * rowlen + 1 * TD_WAVE + lane] : NEG_INF; // :3610
		}
		if (ka.mode == MODE_ARCH_COMP) { // do_arch_comparison, barcode_hmm.c:2111-2148: backward only
			ka.out_b[rid] = b_score;
			continue;
		}
		// A read without any valid path has b_score = -inf; the reference then indexes its table with NaN
		// and crashes.  Here such a read is reported as an architecture mismatch with Q = 0.
		const bool dead = !(b_score > NEG_INF);
		const float b_use = dead ? 0.0f : b_score;

		// ---- phase 2: forward + posteriors, :4155-4349 ----
		float f_score;
		{
			float* prev = SF; // row 0 = previous_silent: [0] = 0, [1..len] = -inf
			for (int i = 0; i <= tmax; i++) prev[i * TD_WAVE + lane] = (i == 0) ? 0.0f : NEG_INF;
			float run = 0.0f;
			for (int j = 0; j < S; j++) {
				run = run + hd.seg[j].skip; // :4172-4176
				SF[(int64_t)(j + 1) * rowlen + lane] = run;
			}
			for (int j = 0; j < S; j++) {
				const TdSeg sg = load_seg(&hd, j);
				const float* P = SF + (int64_t)j * rowlen;
				float* Cs = SF + (int64_t)(j + 1) * rowlen;
				TD_DISPATCH_NCOL(sg.n_col,
					fwd_segment_reg<NC>(w, sg, cols_c, codes, P, Cs, BW, DP, TOTAL, H, b_use),
					for (int f = 0; f < sg.n_hmm; f++) {
						const int col0 = sg.col_off + f * sg.n_col;
						const int h = sg.hmm_off + f;
						const float tot = fwd_hmm_mem(w, sg.n_col, cols_c + col0, codes, P, Cs,
						                              BW + (int64_t)col0 * lmax * TD_WAVE, FWROW,
						                              DP + (int64_t)h * TD_WAVE, H, b_use, f == 0, sg.skip_live != 0, sg.skip);
						TOTAL[h * TD_WAVE + lane] = tot;
					})
			}
			f_score = (len >= 1) ? SF[(int64_t)S * rowlen + len * TD_WAVE + lane] : NEG_INF; // :4349
		}

		// ---- barcode confidence, :4354-4429 ----
		float bar_prob;
		{
			int hc = 0;
			for (int j = 0; j < S; j++) {
				const int n = hd.seg[j].n_hmm;
				if (n > 1) {
					float n1 = NEG_INF;
					for (int f = 0; f < n; f++) n1 = lsum(T, n1, TOTAL[(hc + f) * TD_WAVE + lane]);
					for (int f = 0; f < n; f++) TOTAL[(hc + f) * TD_WAVE + lane] = TOTAL[(hc + f) * TD_WAVE + lane] - n1;
				}
				hc += n;
			}
			hc = 0;
			int gg = 1;
			float n0 = NEG_INF, n2 = 0.0f;
			for (int j = 0; j < S; j++) {
				const int n = hd.seg[j].n_hmm;
				if (n > 1) {
					gg = 0;
					float n1 = NEG_INF;
					for (int f = 0; f < n; f++) {
						const float tv = TOTAL[(hc + f) * TD_WAVE + lane];
						if (tv > n0 && f != n - 1) n0 = tv;
						n1 = lsum(T, n1, tv);
					}
					n0 = n0 - n1; // not reset between segments, :4387
					n2 = n2 + n0;
				}
				hc += n;
			}
			bar_prob = (gg || n2 > 0.0f) ? 0.0f : n2;
		}

		// ---- random model, :4516-4523 ----
		float r_score = 0.0f;
		for (int i = 1; i <= tmax; i++) {
			if (i <= len) r_score = (r_score + emit5(hd.bg, codes[i * TD_WAVE + lane])) + hd.r_stay;
		}
		r_score += hd.r_exit;

		// ---- Q value, do_label_thread :2320-2338 ----
		float Q;
		{
			float pbest = f_score;               // logsum(-inf, f_score) == f_score
			pbest = lsum(T, pbest, r_score);
			const double t = ((double)bar_prob + (double)f_score) - (double)pbest;
			const float e = post_prob((float)t);
			pbest = (float)(1.0 - (double)e);
			if (pbest == 0.0f) Q = 40.0f;
			else if (pbest == 1.0f) Q = 0.0f;
			else Q = (float)(-10.0 * log10((double)pbest));
		}
		if (dead || len < 1) Q = 0.0f;

		int read_type = 0, barcode = -1, fingerprint = -1;
		const int nw1 = ka.nw1;
		uint32_t* keep = ka.out_keep + (int64_t)tile * nw1 * TD_WAVE;

		if (ka.mode == MODE_GET_LABEL) {
			int8_t* labels = ka.out_labels + (int64_t)tile * (lmax + 1) * TD_WAVE;
			// ---- label DP, :4447-4472 (acc = dyn_prog_matrix row, updated in place from the highest label down) ----
			for (int v = 0; v < H; v++) ACC[v * TD_WAVE + lane] = 0.0f; // row 0: scaledprob2prob(-inf)
			for (int i = 1; i <= tmax; i++) {
				if (i <= len) {
					const float* dpi = DP + (int64_t)(i - 1) * H * TD_WAVE;
					uint8_t* pathi = PATH + (int64_t)(i - 1) * H * TD_WAVE;
					for (int v = H - 1; v >= 0; v--) {
						float m = -1.0f;
						int mv = 0;
						const int p0 = pred_off[v], p1 = pred_off[v + 1];
						for (int p = p0; p < p1; p++) {      // predecessors u < v in ascending order
							const int u = pred_idx[p];
							const float au = ACC[u * TD_WAVE + lane];
							if (au > m) { m = au; mv = u; }
						}
						const float self = ACC[v * TD_WAVE + lane];
						if (self >= m) { m = self; mv = v; } // staying in v wins ties (:4461-4464)
						ACC[v * TD_WAVE + lane] = dpi[v * TD_WAVE + lane] + m;
						pathi[v * TD_WAVE + lane] = (uint8_t)mv;
					}
				}
			}
			// termination + traceback, :4494-4514
			{
				float m = -1.0f;
				int move = 0;
				for (int v = 0; v < H; v++) {
					const float a = ACC[v * TD_WAVE + lane];
					if (a > m) { m = a; move = v; }
				}
				if (len >= 1) labels[len * TD_WAVE + lane] = (int8_t)move;
				for (int i = tmax; i >= 1; i--) {
					if (i <= len) {
						move = PATH[((int64_t)(i - 1) * H + move) * TD_WAVE + lane];
						labels[(i - 1) * TD_WAVE + lane] = (int8_t)move;
					}
				}
			}

			// with a window, ri->labels beyond it keeps the zeros read_fasta_fastq() put there (io.c:1755-1764)
			if (ka.win_len > 0) {
				for (int i = 1; i <= tmaxF; i++)
					if (i > len && i <= lenF) labels[i * TD_WAVE + lane] = 0;
			}
			// ---- extract_reads, :3172-3313 ----
			bool extracted = false;
			for (int k = 0; k < nw1; k++) keep[k * TD_WAVE + lane] = 0xFFFFFFFFu;
			if (dead || len < 1 || !(ka.threshold <= Q)) {
				read_type = OUT_ARCH_MISMATCH;
			} else {
				uint32_t key = 0;
				int bar = -1, mem = -1, fingerlen = 0, s_pos = 0, has_bar = 0;
				bool too_short = false, in_read = false, stopped = false;
				uint32_t kw = 0;
				// make_extracted_read (:3336-3352) walks the whole read with labels[j+1] on position j: beyond a window those
				// labels are 0, i.e. HMM 0 of segment 0 -- kept only if that segment is a read segment
				const bool tail_kept = (ka.hinfo[0] & 0xFF) == 'R';
				const int xmax = ka.win_len > 0 ? tmaxF : tmax;
				for (int jx = 0; jx < xmax; jx++) {
					if (jx < len) {
						const uint32_t hi = ka.hinfo[(int)labels[(jx + 1) * TD_WAVE + lane]];
						const int ty = hi & 0xFF;
						if (!stopped) {
							if (ty == 'F') {
								fingerlen++;
								key = (key << 2) | (uint32_t)(codes[(jx + 1) * TD_WAVE + lane] & 3);
							}
							if (ty == 'B') {
								has_bar = (hi >> 31) ? -1 : 1;
								bar = (hi >> 16) & 0x7FFF;
								mem = (hi >> 8) & 0xFF;
							}
							if (ty == 'R') {
								s_pos++;
								in_read = true;
							} else {
								if (in_read && s_pos < ka.minlen) { too_short = true; stopped = true; }
								in_read = false;
								s_pos = 0;
							}
						}
						if (ty == 'R') kw |= 1u << (jx & 31);
					} else if (jx < lenF && tail_kept) {
						kw |= 1u << (jx & 31);
					}
					if ((jx & 31) == 31 || jx == xmax - 1) { keep[(jx >> 5) * TD_WAVE + lane] = kw; kw = 0; }
				}
				if (in_read && s_pos < ka.minlen) too_short = true;
				const int req = hd.required_finger_len;
				const int fp = (int)((key << 8) | (uint32_t)(req <= 255 ? req : 255));
				if (too_short) {
					read_type = OUT_TOO_SHORT;
				} else if (has_bar == -1) {
					read_type = OUT_BAR_FINGER_NOT_FOUND;
				} else if (has_bar && req) {
					if (fingerlen == req && bar != -1) { extracted = true; barcode = (mem << 16) | bar; fingerprint = fp; }
					else read_type = OUT_BAR_FINGER_NOT_FOUND;
				} else if (has_bar) {
					if (bar != -1) { extracted = true; barcode = (mem << 16) | bar; }
					else read_type = OUT_BAR_FINGER_NOT_FOUND;
				} else if (req) {
					if (fingerlen == req) { extracted = true; fingerprint = fp; }
					else read_type = OUT_BAR_FINGER_NOT_FOUND;
				} else {
					extracted = true;
				}
				if (extracted) read_type = OUT_SUCCESS;
			}
			// keep mask = positions whose byte the reference leaves untouched; all ones unless make_extracted_read ran
			if (!extracted) for (int k = 0; k < nw1; k++) keep[k * TD_WAVE + lane] = 0xFFFFFFFFu;

			// ---- match_to_reference, :2478-2583 (td_artifact.inc) ----
			if (ka.art_n > 0) {
				const uint8_t* cd = codesF;
				auto sq = [&](int kk) -> int {
					return ((keep[(kk >> 5) * TD_WAVE + lane] >> (kk & 31)) & 1u) ? (int)cd[(kk + 1) * TD_WAVE + lane] : 65;
				};
				const int id = td_art_match(sq, lenF, tmaxF, ka.art_left[rid] != 0, ka.art_text, ka.art_index, ka.art_n, ka.art_fe);
				if (id > 0 && read_type == OUT_SUCCESS) read_type = (id << 8) | OUT_MATCHES_ARTIFACTS;
			}

			// ---- dust_sequences, :2407-2467, on the rewritten sequence ----
			if (ka.dust && lenF >= 1) {
#define SQ(kk) (((kk) < lenF) ? ((((keep[((kk) >> 5) * TD_WAVE + lane] >> ((kk) & 31)) & 1u)) ? (int)codesF[((kk) + 1) * TD_WAVE + lane] : 65) : 0)
				for (int k = 0; k < 64; k++) DUST[k * TD_WAVE + lane] = 0;
				int c0 = 0;
				while (SQ(c0) == 65) c0++;
				int key = ((SQ(c0) & 3) << 2) | (SQ(c0 + 1) & 3);
				const int n = lenF > 64 ? 64 : lenF;
				int cc = c0 + 2;
				for (int jx = c0 + 2; jx < n; jx++) {
					const int s = SQ(jx);
					if (s == 65) break;
					key = (int)(((uint32_t)key << 2) | (uint32_t)(s & 3));
					DUST[(key & 63) * TD_WAVE + lane] += 1;
					cc++;
				}
				double s = 0.0;
				for (int k = 0; k < 64; k++) {
					const double tcount = (double)DUST[k * TD_WAVE + lane];
					s += tcount * (tcount - 1.0) / 2.0;
				}
				s = s / (double)(cc - 3) * 10.0;
				if (s > (double)ka.dust) read_type = OUT_LOW_COMPLEXITY;
#undef SQ
			}
		}

		// ---- per-read outputs + counters ----
		if (lenF >= 1 || ka.lens[rid] == 0) {
			ka.out_f[rid] = f_score;
			ka.out_b[rid] = b_score;
			ka.out_r[rid] = r_score;
			ka.out_bar[rid] = bar_prob;
			ka.out_q[rid] = Q;
			ka.out_type[rid] = read_type;
			ka.out_barcode[rid] = barcode;
			ka.out_finger[rid] = fingerprint;
		}
		if (ka.mode == MODE_GET_LABEL) {
			// one atomic per wave and distinct value (see td_spec_kernel.inc: a million atomics on two or three words hold up
			// every memory operation of the machine)
			const int key1 = (lenF >= 1) ? (read_type & (N_OUTCOME_SLOTS - 1)) : -1;
			const int key2 = (lenF >= 1 && read_type == OUT_SUCCESS && barcode >= 0) ? N_OUTCOME_SLOTS + (barcode & 0xFF) : -1;
#pragma unroll
			for (int pass = 0; pass < 2; pass++) {
				const int key = pass ? key2 : key1;
				unsigned long long todo = __builtin_amdgcn_ballot_w64(key >= 0);
				while (todo) {
					const int leader = __builtin_ctzll(todo);
					const int kv = __builtin_amdgcn_readlane(key, leader);
					const unsigned long long same = __builtin_amdgcn_ballot_w64(key == kv);
					if (lane == leader) atomicAdd(&ka.counters[kv], (unsigned long long)__builtin_popcountll(same));
					todo &= ~same;
				}
			}
		}
	}
}

__global__ __launch_bounds__(TD_BLOCK, 2) void td_decode_kernel(const TdKernelArgs ka)
{
	decode_body(ka, blockIdx.x);
}

// Many models over one batch in one launch (architecture comparison, test_architectures.c:182-184 calls run_pHMM with all
// candidate model bags): blockIdx.y selects the model's argument block -- its own tables, workspace region and output.
__global__ __launch_bounds__(TD_BLOCK, 2) void td_decode_multi_kernel(const TdKernelArgs* __restrict__ args)
{
	const TdKernelArgs ka = args[blockIdx.y];
	decode_body(ka, blockIdx.x);
}

extern "C" __attribute__((visibility("hidden"))) hipError_t td_launch_decode_multi(const TdKernelArgs* d_args, int n_models, int max_slots, hipStream_t stream)
{
	const int blocks = (max_slots + TD_WAVES_PER_BLOCK - 1) / TD_WAVES_PER_BLOCK;
	hipLaunchKernelGGL(td_decode_multi_kernel, dim3(blocks, n_models), dim3(TD_BLOCK), 0, stream, d_args);
	return hipGetLastError();
}

extern "C" __attribute__((visibility("hidden"))) hipError_t td_launch_decode(const TdKernelArgs* ka, hipStream_t stream)
{
	const int blocks = (ka->n_slots + TD_WAVES_PER_BLOCK - 1) / TD_WAVES_PER_BLOCK;
	hipLaunchKernelGGL(td_decode_kernel, dim3(blocks), dim3(TD_BLOCK), 0, stream, *ka);
	return hipGetLastError();
}

extern "C" __attribute__((visibility("hidden"))) int td_kernel_block_threads(void) { return TD_BLOCK; }
