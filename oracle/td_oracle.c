/*
 * td_oracle.c -- TEST INFRASTRUCTURE ONLY (parity checker + bench.py cpu_baseline "port").
 *
 * Plain-C CPU restatement, on flattened tables, of TagDust2's per-read HMM decoding path.
 * Own code; every function cites the reference file:line (relative to /root/reference/src) it follows.
 * The arithmetic contract is IEEE float32, evaluated in the reference's order, with logsum() as the
 * 16000-entry table operation -- so this file must be built without FMA contraction
 * (-ffp-contract=off, see Makefile).
 *
 * Parity status: PINNED (see td_oracle.h).
 */
#include "td_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#define NEG_INF (-INFINITY)

/* ------------------------------------------------------------------------------------------------
 * log-space primitives, misc.c:57-105
 * ---------------------------------------------------------------------------------------------- */
static float g_logsum_table[TDO_LOGSUM_SIZE];
static int g_logsum_ready = 0;

void tdo_init_logsum(void)
{
	/* misc.c:57-63: logsum_lookup[i] = log(1. + exp((double) -i / SCALE)), SCALE = 1000.0f */
	for (int i = 0; i < TDO_LOGSUM_SIZE; i++) {
		g_logsum_table[i] = (float)log(1.0 + exp((double)-i / (double)1000.0f));
	}
	g_logsum_ready = 1;
}

const float* tdo_logsum_table(void)
{
	if (!g_logsum_ready) tdo_init_logsum();
	return g_logsum_table;
}

static inline float logsum_(float a, float b)
{
	/* misc.c:72-78 */
	const float mx = (a > b) ? a : b;
	const float mn = (a < b) ? a : b;
	if (mn == NEG_INF || (mx - mn) >= 15.7f) return mx;
	return mx + g_logsum_table[(int)((mx - mn) * 1000.0f)];
}
float tdo_logsum(float a, float b) { return logsum_(a, b); }
#define LS(a, b) logsum_((a), (b))

/* scaledprob2prob(), misc.c:98-105: float in, exp() in double, float out */
static float sp2p(float p)
{
	if (p == NEG_INF) return 0.0f;
	return (float)exp((double)p);
}
/* prob2scaledprob(), misc.c:85-92: float in, log() in double, float out */
static float p2sp(float p)
{
	if (p == 0.0f) return NEG_INF;
	return (float)log((double)p);
}

/* ------------------------------------------------------------------------------------------------
 * workspace: the DP rows the reference keeps inside struct hmm_column / struct model / model_bag
 * ---------------------------------------------------------------------------------------------- */
struct tdo_workspace {
	int C, S, H, stride; /* stride >= max_len + 2 */
	float *MB, *IB, *DB; /* [C][stride] */
	float *MF, *IF, *DF; /* [C][stride] */
	float *SB, *SF;      /* [S][stride] */
	float *prev;         /* [stride] previous_silent == next_silent (barcode_hmm.c:4151-4152) */
	float *dp;           /* [stride][H] dyn_prog_matrix */
	int   *path;         /* [stride][H] */
};

tdo_workspace* tdo_workspace_new(const tdo_model* m, int max_len)
{
	tdo_workspace* ws = (tdo_workspace*)calloc(1, sizeof(*ws));
	if (!ws) return NULL;
	ws->C = m->C; ws->S = m->S; ws->H = m->H; ws->stride = max_len + 2;
	size_t cs = (size_t)m->C * ws->stride, ss = (size_t)m->S * ws->stride;
	ws->MB = (float*)malloc(sizeof(float) * cs); ws->IB = (float*)malloc(sizeof(float) * cs);
	ws->DB = (float*)malloc(sizeof(float) * cs); ws->MF = (float*)malloc(sizeof(float) * cs);
	ws->IF = (float*)malloc(sizeof(float) * cs); ws->DF = (float*)malloc(sizeof(float) * cs);
	ws->SB = (float*)malloc(sizeof(float) * ss); ws->SF = (float*)malloc(sizeof(float) * ss);
	ws->prev = (float*)malloc(sizeof(float) * ws->stride);
	ws->dp = (float*)malloc(sizeof(float) * (size_t)ws->stride * m->H);
	ws->path = (int*)malloc(sizeof(int) * (size_t)ws->stride * m->H);
	if (!ws->MB || !ws->IB || !ws->DB || !ws->MF || !ws->IF || !ws->DF || !ws->SB || !ws->SF ||
	    !ws->prev || !ws->dp || !ws->path) {
		tdo_workspace_free(ws);
		return NULL;
	}
	return ws;
}

void tdo_workspace_free(tdo_workspace* ws)
{
	if (!ws) return;
	free(ws->MB); free(ws->IB); free(ws->DB); free(ws->MF); free(ws->IF); free(ws->DF);
	free(ws->SB); free(ws->SF); free(ws->prev); free(ws->dp); free(ws->path);
	free(ws);
}

/* base code x_i for 1-based position i; the reference reads a[len] == 0 (the loader's terminator,
 * io.c:1759) for i == len+1 */
static inline int base_at(const uint8_t* seq, int len, int i)
{
	return (i >= 1 && i <= len) ? (int)seq[i - 1] : 0;
}

/* ------------------------------------------------------------------------------------------------
 * backward(), barcode_hmm.c:3439-3640
 * ---------------------------------------------------------------------------------------------- */
float tdo_backward(const tdo_model* m, tdo_workspace* ws, const uint8_t* seq, int len)
{
	const int st = ws->stride;
	const int S = m->S;
	int i, j, f, g;

	/* init, :3466-3492 */
	for (int c = 0; c < m->C; c++) {
		for (i = 0; i <= len + 1; i++) {
			ws->MB[c * st + i] = NEG_INF;
			ws->IB[c * st + i] = NEG_INF;
			ws->DB[c * st + i] = NEG_INF;
		}
	}
	for (j = 0; j < S; j++)
		for (i = 0; i <= len + 1; i++) ws->SB[j * st + i] = NEG_INF;
	for (i = 0; i <= len + 1; i++) ws->prev[i] = NEG_INF;
	ws->prev[len + 1] = 0.0f;
	ws->SB[(S - 1) * st + len + 1] = 0.0f + m->skip[S - 1];
	for (j = S - 2; j >= 0; j--) ws->SB[j * st + len + 1] = ws->SB[(j + 1) * st + len + 1] + m->skip[j];

	/* :3496-3608 */
	for (j = S - 1; j >= 0; j--) {
		const float* P = (j == S - 1) ? ws->prev : &ws->SB[(j + 1) * st];
		float* Cs = &ws->SB[j * st];
		const int K = m->n_col[j] - 1;
		for (f = 0; f < m->n_hmm[j]; f++) {
			const int base = m->col_off[j] + f * m->n_col[j];
			for (i = len; i > 0; i--) {
				const int c = base_at(seq, len, i + 1);
				const int xi = base_at(seq, len, i);
				/* last column, :3518-3543 */
				{
					const int k = base + K;
					const float* t = &m->trans[k * 9];
					const float* em = &m->eM[k * 5];
					const float* ei = &m->eI[k * 5];
					float* M = &ws->MB[k * st];
					float* I = &ws->IB[k * st];
					M[i] = P[i + 1] + t[TDO_MSKIP];
					I[i] = P[i + 1] + t[TDO_ISKIP];
					I[i] = LS(I[i], M[i + 1] + t[TDO_IM] + em[c]);
					I[i] = LS(I[i], I[i + 1] + t[TDO_II] + ei[c]);
					Cs[i] = LS(Cs[i], M[i] + m->sM[k] + em[xi]);
					Cs[i] = LS(Cs[i], I[i] + m->sI[k] + ei[xi]);
					ws->DB[k * st + i] = NEG_INF;
				}
				/* :3544-3586 */
				for (g = K - 1; g >= 0; g--) {
					const int k = base + g, p = k + 1;
					const float* t = &m->trans[k * 9];
					const float* em = &m->eM[k * 5];
					const float* ei = &m->eI[k * 5];
					const float* pem = &m->eM[p * 5];
					float* M = &ws->MB[k * st];
					float* I = &ws->IB[k * st];
					float* D = &ws->DB[k * st];
					const float* pM = &ws->MB[p * st];
					const float* pD = &ws->DB[p * st];

					M[i] = pM[i + 1] + pem[c] + t[TDO_MM];
					M[i] = LS(M[i], P[i + 1] + t[TDO_MSKIP]);
					M[i] = LS(M[i], I[i + 1] + ei[c] + t[TDO_MI]);
					M[i] = LS(M[i], pD[i] + t[TDO_MD]);

					I[i] = I[i + 1] + t[TDO_II] + ei[c];
					I[i] = LS(I[i], P[i + 1] + t[TDO_ISKIP]);
					I[i] = LS(I[i], pM[i + 1] + t[TDO_IM] + pem[c]);

					D[i] = pD[i] + t[TDO_DD];
					D[i] = LS(D[i], pM[i] + pem[xi] + t[TDO_DM]);

					Cs[i] = LS(Cs[i], M[i] + m->sM[k] + em[xi]);
					Cs[i] = LS(Cs[i], I[i] + m->sI[k] + ei[xi]);
				}
				/* :3600 -- inside the per-HMM loop (quirk: applied n_hmm times) */
				Cs[i] = LS(Cs[i], P[i] + m->skip[j]);
			}
		}
	}
	return ws->SB[0 * st + 1]; /* :3610 */
}

/* ------------------------------------------------------------------------------------------------
 * forward_max_posterior_decoding(), barcode_hmm.c:4128-4525
 * ---------------------------------------------------------------------------------------------- */
void tdo_forward_decode(const tdo_model* m, tdo_workspace* ws, const uint8_t* seq, int len,
                        float b, float* f_score, float* r_score, float* bar_prob, int8_t* labels)
{
	const int st = ws->stride;
	const int S = m->S, H = m->H;
	int i, j, f, g, h;
	float total[128];

	/* init, :4155-4196 */
	for (int c = 0; c < m->C; c++) {
		for (i = 0; i <= len; i++) {
			ws->MF[c * st + i] = NEG_INF;
			ws->IF[c * st + i] = NEG_INF;
			ws->DF[c * st + i] = NEG_INF;
		}
	}
	for (j = 0; j < S; j++)
		for (i = 0; i <= len + 1; i++) ws->SF[j * st + i] = NEG_INF;
	ws->SF[0] = 0.0f + m->skip[0];
	for (j = 1; j < S; j++) ws->SF[j * st] = ws->SF[(j - 1) * st] + m->skip[j];
	for (i = 0; i <= len; i++) {
		for (h = 0; h < H; h++) {
			ws->dp[i * H + h] = NEG_INF;
			ws->path[i * H + h] = -1;
		}
	}
	for (h = 0; h < H; h++) total[h] = NEG_INF;
	for (i = 0; i <= len; i++) ws->prev[i] = NEG_INF;
	ws->prev[0] = 0.0f;
	ws->prev[len + 1] = 0.0f;

	/* :4201-4347 */
	h = 0;
	for (j = 0; j < S; j++) {
		const float* P = (j == 0) ? ws->prev : &ws->SF[(j - 1) * st];
		float* Cs = &ws->SF[j * st];
		const int ncol = m->n_col[j];
		for (f = 0; f < m->n_hmm[j]; f++) {
			const int base = m->col_off[j] + f * ncol;
			for (i = 1; i <= len; i++) {
				const int c = base_at(seq, len, i);
				float* dpi = &ws->dp[i * H + h];
				/* column 0, :4220-4268 */
				{
					const int k = base;
					const float* t = &m->trans[k * 9];
					float* M = &ws->MF[k * st];
					float* I = &ws->IF[k * st];
					M[i] = P[i - 1] + m->sM[k] + m->eM[k * 5 + c];
					total[h] = LS(total[h], M[i] + ws->MB[k * st + i] - b);
					*dpi = LS(*dpi, M[i] + ws->MB[k * st + i] - b);

					I[i] = P[i - 1] + m->sI[k];
					I[i] = LS(I[i], I[i - 1] + t[TDO_II]);
					I[i] = LS(I[i], M[i - 1] + t[TDO_MI]);
					I[i] = I[i] + m->eI[k * 5 + c];

					total[h] = LS(total[h], P[i - 1] + m->sI[k] + m->eI[k * 5 + c] + ws->IB[k * st + i] - b);
					*dpi = LS(*dpi, I[i] + ws->IB[k * st + i] - b);

					ws->DF[k * st + i] = NEG_INF;
					Cs[i] = LS(Cs[i], M[i] + t[TDO_MSKIP]);
					Cs[i] = LS(Cs[i], I[i] + t[TDO_ISKIP]);
				}
				/* :4271-4334 */
				for (g = 1; g < ncol; g++) {
					const int k = base + g, p = k - 1;
					const float* t = &m->trans[k * 9];
					const float* pt = &m->trans[p * 9];
					float* M = &ws->MF[k * st];
					float* I = &ws->IF[k * st];
					float* D = &ws->DF[k * st];
					const float* pM = &ws->MF[p * st];
					const float* pI = &ws->IF[p * st];
					const float* pD = &ws->DF[p * st];

					M[i] = P[i - 1] + m->sM[k];
					M[i] = LS(M[i], pM[i - 1] + pt[TDO_MM]);
					M[i] = LS(M[i], pI[i - 1] + pt[TDO_IM]);
					M[i] = LS(M[i], pD[i] + pt[TDO_DM]);
					M[i] = M[i] + m->eM[k * 5 + c];
					*dpi = LS(*dpi, M[i] + ws->MB[k * st + i] - b);

					I[i] = P[i - 1] + m->sI[k];
					I[i] = LS(I[i], I[i - 1] + t[TDO_II]);
					I[i] = LS(I[i], M[i - 1] + t[TDO_MI]);
					I[i] = I[i] + m->eI[k * 5 + c];
					*dpi = LS(*dpi, I[i] + ws->IB[k * st + i] - b);

					D[i] = pM[i] + pt[TDO_MD];
					D[i] = LS(D[i], pD[i] + pt[TDO_DD]);
					/* :4325-4327 write only the training estimates transition_e[] -- dead for labelling */

					Cs[i] = LS(Cs[i], M[i] + t[TDO_MSKIP]);
					Cs[i] = LS(Cs[i], I[i] + t[TDO_ISKIP]);
				}
				Cs[i] = LS(Cs[i], P[i] + m->skip[j]); /* :4341 */
			}
			h++;
		}
	}
	*f_score = ws->SF[(S - 1) * st + len]; /* :4349 */

	/* barcode confidence, :4354-4429 (scratch scalars n0,n1,n2 are next_silent[0..2]) */
	{
		float n0, n1, n2;
		int hc = 0, gg;
		for (j = 0; j < S; j++) {
			if (m->n_hmm[j] > 1) {
				gg = hc;
				n1 = NEG_INF;
				for (f = 0; f < m->n_hmm[j]; f++) { n1 = LS(n1, total[hc]); hc++; }
				for (f = 0; f < m->n_hmm[j]; f++) { total[gg] = total[gg] - n1; gg++; }
			} else {
				hc += m->n_hmm[j];
			}
		}
		hc = 0; gg = 1;
		n0 = NEG_INF; n1 = NEG_INF; n2 = 0.0f;
		for (j = 0; j < S; j++) {
			if (m->n_hmm[j] > 1) {
				gg = 0;
				n1 = NEG_INF;
				for (f = 0; f < m->n_hmm[j]; f++) {
					if (total[hc] > n0 && f != m->n_hmm[j] - 1) n0 = total[hc];
					n1 = LS(n1, total[hc]);
					hc++;
				}
				n0 = n0 - n1; /* n0 is NOT reset between segments (:4387 vs the commented :4393) */
				n2 = n2 + n0;
			} else {
				hc += m->n_hmm[j];
			}
		}
		if (gg) *bar_prob = 0.0f;
		else if (n2 > 0) *bar_prob = 0.0f;
		else *bar_prob = n2;
	}

	/* posteriors -> probabilities, :4431-4440 */
	for (i = 0; i <= len; i++)
		for (h = 0; h < H; h++) ws->dp[i * H + h] = sp2p(ws->dp[i * H + h]);

	/* label DP, :4447-4472 */
	{
		float mx, tmp;
		int move = -1, c;
		for (i = 1; i <= len; i++) {
			for (j = 0; j < H; j++) {
				mx = -1;
				for (c = 0; c <= j; c++) {
					tmp = ws->dp[(i - 1) * H + c] * m->A[c * H + j];
					if (tmp > mx) { move = c; mx = tmp; }
					if (tmp == mx && c == j) { move = c; mx = tmp; }
				}
				ws->dp[i * H + j] += mx;
				ws->path[i * H + j] = move;
			}
		}
		/* termination + traceback, :4494-4514 */
		mx = -1;
		for (j = 0; j < H; j++) {
			if (ws->dp[len * H + j] > mx) { mx = ws->dp[len * H + j]; move = j; }
		}
		for (i = 0; i <= len; i++) labels[i] = 0;
		labels[len] = (int8_t)move;
		for (i = len; i > 0; i--) {
			move = ws->path[i * H + move];
			labels[i - 1] = (int8_t)move;
		}
	}

	/* random model, :4516-4523 */
	{
		float r = 0.0f;
		const float stay = p2sp((float)(1.0 - (1.0 / (double)(float)m->avg_len)));
		for (i = 1; i <= len; i++) r = r + m->bg[base_at(seq, len, i)] + stay;
		r += p2sp((float)(1.0 / (double)(float)m->avg_len));
		*r_score = r;
	}
}

/* ------------------------------------------------------------------------------------------------
 * Q value, do_label_thread barcode_hmm.c:2320-2338
 * ---------------------------------------------------------------------------------------------- */
float tdo_qvalue(float f_score, float r_score, float bar_prob)
{
	float pbest = NEG_INF; /* ri->mapq = prob2scaledprob(0.0), :2287 */
	pbest = LS(pbest, f_score);
	pbest = LS(pbest, r_score);
	/* ri->bar_prob is a double field (io.h:86): the subtraction runs in double and is narrowed to the
	 * float parameter of scaledprob2prob(); 1.0 - float runs in double and is narrowed to float pbest */
	{
		const double t = ((double)bar_prob + (double)f_score) - (double)pbest;
		pbest = (float)(1.0 - (double)sp2p((float)t));
	}
	if (!pbest) return 40.0f;
	if (pbest == 1.0) return 0.0f;
	return (float)(-10.0 * log10((double)pbest));
}

/* ------------------------------------------------------------------------------------------------
 * extract_reads + make_extracted_read, barcode_hmm.c:3172-3356
 * ---------------------------------------------------------------------------------------------- */
static void make_extracted(const tdo_model* m, uint8_t* seq, uint8_t* qual, int len, const int8_t* labels)
{
	/* :3325-3356: s_pos advances on both branches, so R positions keep their place and every other
	 * position becomes the spacer byte 65 */
	for (int j = 0; j < len; j++) {
		const int seg = m->label[(int)labels[j + 1]] & 0xFFFF;
		if (m->type[seg] != 'R') {
			seq[j] = 65;
			if (qual) qual[j] = 65;
		}
	}
}

void tdo_extract(const tdo_model* m, const tdo_params* p, uint8_t* seq, uint8_t* qual, int len,
                 const int8_t* labels, float Q, int32_t* read_type, int32_t* barcode, int32_t* fingerprint)
{
	tdo_extract_window(m, p, seq, qual, len, 0, len, labels, Q, read_type, barcode, fingerprint);
}

void tdo_extract_window(const tdo_model* m, const tdo_params* p, uint8_t* seq, uint8_t* qual, int len, int woff, int wlen,
                        const int8_t* labels, float Q, int32_t* read_type, int32_t* barcode, int32_t* fingerprint)
{
	int j, c1, c2, c3;
	uint32_t key = 0;
	int bar = -1, mem = -1, fingerlen = 0, required = 0;
	int s_pos = 0, has_bar = 0, too_short = 0, in_read = 0;

	for (j = 0; j < m->S; j++)
		if (m->type[j] == 'F') required += m->finger_len[j];

	if (!(p->threshold <= Q)) { /* :3203 / :3301-3305 */
		*read_type = TDO_FAIL_ARCHITECTURE_MISMATCH;
		return;
	}
	for (j = 0; j < wlen; j++) { /* :3205-3242; with a window len = matchend - matchstart, offset = matchstart (:3189-3193) */
		c1 = m->label[(int)labels[j + 1]];
		c2 = c1 & 0xFFFF;
		c3 = (c1 >> 16) & 0x7FFF;
		if (m->type[c2] == 'F') {
			fingerlen++;
			key = (key << 2) | (uint32_t)(seq[j + woff] & 0x3);
		}
		if (m->type[c2] == 'B') {
			has_bar = 1;
			bar = c3;
			if (bar == m->n_hmm[c2] - 1) has_bar = -1; /* the all-N decoy, interface.c:521-527 */
			mem = c2;
		}
		if (m->type[c2] == 'R') {
			s_pos++;
			in_read = 1;
		} else {
			if (in_read && s_pos < p->minlen) { too_short = 1; break; }
			in_read = 0;
			s_pos = 0;
		}
	}
	if (in_read && s_pos < p->minlen) too_short = 1;

	if (too_short) { *read_type = TDO_FAIL_READ_TOO_SHORT; return; }
	{
		const int32_t fp = (int32_t)((key << 8) | (uint32_t)(required <= 255 ? required : 255));
		if (has_bar == -1) {
			*read_type = TDO_FAIL_BAR_FINGER_NOT_FOUND;
		} else if (has_bar && required) {
			if (fingerlen == required && bar != -1) {
				make_extracted(m, seq, qual, len, labels);
				*barcode = (mem << 16) | bar;
				*fingerprint = fp;
				*read_type = TDO_EXTRACT_SUCCESS;
			} else {
				*read_type = TDO_FAIL_BAR_FINGER_NOT_FOUND;
			}
		} else if (has_bar) {
			if (bar != -1) {
				make_extracted(m, seq, qual, len, labels);
				*barcode = (mem << 16) | bar;
				*read_type = TDO_EXTRACT_SUCCESS;
			} else {
				*read_type = TDO_FAIL_BAR_FINGER_NOT_FOUND;
			}
		} else if (required) {
			if (fingerlen == required) {
				make_extracted(m, seq, qual, len, labels);
				*fingerprint = fp;
				*read_type = TDO_EXTRACT_SUCCESS;
			} else {
				*read_type = TDO_FAIL_BAR_FINGER_NOT_FOUND;
			}
		} else {
			make_extracted(m, seq, qual, len, labels);
			*read_type = TDO_EXTRACT_SUCCESS;
		}
	}
}

/* ------------------------------------------------------------------------------------------------
 * dust_sequences, barcode_hmm.c:2407-2467 (one read)
 * ---------------------------------------------------------------------------------------------- */
int tdo_dust(const uint8_t* seq, int len, int dust_cut)
{
#define SEQ_AT(k) (((k) < len) ? seq[(k)] : 0) /* the reference's seq carries a 0 terminator */
	double triplet[64];
	double s = 0.0;
	int c = 0, j, key, n;
	for (j = 0; j < 64; j++) triplet[j] = 0.0;
	while (SEQ_AT(c) == 65) c++;
	key = ((SEQ_AT(c) & 0x3) << 2) | (SEQ_AT(c + 1) & 0x3);
	n = len > 64 ? 64 : len;
	c += 2;
	for (j = c; j < n; j++) {
		if (seq[j] == 65) break;
		key = (int)(((uint32_t)key << 2) | (uint32_t)(seq[j] & 0x3));
		triplet[key & 0x3F]++;
		c++;
	}
	for (j = 0; j < 64; j++) s += triplet[j] * (triplet[j] - 1.0) / 2.0;
	s = s / (double)(c - 3) * 10.0;
	return s > dust_cut;
#undef SEQ_AT
}

/* ------------------------------------------------------------------------------------------------
 * artifact matching: bmp_single misc.c:718-765, bpm_check_error misc.c:581-640, reverse_complement
 * misc.c:827-851, match_to_reference barcode_hmm.c:2478-2583
 * ---------------------------------------------------------------------------------------------- */
static int bmp_single_(const uint8_t* t, const uint8_t* p, int n, int m)
{
	uint64_t VP, VN, D0, HN, HP, X, MASK, B[4] = { 0, 0, 0, 0 };
	int64_t diff;
	int k;
	if (m > 63) m = 63;
	diff = m;
	k = m;
	for (int i = 0; i < m; i++)
		if (p[i] != 65) B[p[i] & 0x3u] |= UINT64_C(1) << i;
	VP = (UINT64_C(1) << m) - 1;
	VN = 0;
	m--;
	MASK = UINT64_C(1) << m;
	for (int i = 0; i < n; i++) {
		X = B[t[i] & 0x3u] | VN;
		D0 = ((VP + (X & VP)) ^ VP) | X;
		HN = VP & D0;
		HP = VN | ~(VP | D0);
		X = HP << 1;
		VN = X & D0;
		VP = (HN << 1) | ~(X | D0);
		diff += (HP & MASK) ? 1 : 0;
		diff -= (HN & MASK) ? 1 : 0;
		if (diff < k) k = (int)diff;
	}
	return k;
}

/* The reference shifts by counts >= 64 (reads longer than 64) and by -1 (no usable base); both are taken modulo 64 as
 * the x86-64 shift instructions the reference's build emits do. */
static int bpm_check_error_(const uint8_t* t, const uint8_t* p, int n, int m)
{
	uint64_t VP, VN, D0, HN, HP, X, MASK, diff, k, B[4] = { 0, 0, 0, 0 };
	int new_len = 0;
	diff = (uint64_t)m;
	for (int i = 0; i < m; i++)
		if (p[i] != 65) { B[p[i] & 0x3] |= UINT64_C(1) << (i & 63); new_len++; }
	if (new_len > 31) new_len = 31;
	m = new_len;
	k = (uint64_t)new_len;
	VP = UINT64_MAX;
	VN = 0;
	m--;
	MASK = UINT64_C(1) << (m & 63);
	for (int i = 0; i < n; i++) {
		X = B[t[i] & 0x3] | VN;
		D0 = ((VP + (X & VP)) ^ VP) | X;
		HN = VP & D0;
		HP = VN | ~(VP | D0);
		X = HP << 1;
		VN = X & D0;
		VP = (HN << 1) | ~(X | D0);
		diff += (HP & MASK) >> (m & 63);
		diff -= (HN & MASK) >> (m & 63);
		if (diff < k) k = diff;
	}
	return (int)k;
}

static void revcomp_(uint8_t* p, int len)
{
	static const uint8_t rev[5] = { 3, 2, 1, 0, 4 }; /* rev_nuc_code, nuc_code.c:68-72 */
	for (int i = 0, j = len - 1; i < j; i++, j--) { const uint8_t x = p[i]; p[i] = p[j]; p[j] = x; }
	for (int i = 0; i < len; i++) if (p[i] != 65) p[i] = rev[p[i] > 4 ? 4 : p[i]];
}

void tdo_match_artifacts(const tdo_artifacts* a, uint8_t* seqs, const int64_t* offs, tdo_result* res,
                         int64_t start, int64_t end)
{
	int64_t i;
	for (i = start; i <= end - 4; i += 4) {
		int errors[4], id[4];
		for (int c = 0; c < 4; c++) { errors[c] = 100000; id[c] = 0; }
		for (int j = 0; j < a->n_seq; j++) {
			const uint8_t* t = a->string + a->s_index[j];
			const int n = a->s_index[j + 1] - a->s_index[j];
			for (int strand = 0; strand < 2; strand++) {
				for (int c = 0; c < 4; c++) {
					uint8_t* q = seqs + offs[i + c];
					const int len = (int)(offs[i + c + 1] - offs[i + c]);
					if (strand) revcomp_(q, len);
					const int e = len > 0 ? bmp_single_(t, q, n, len) : n; /* validate_bpm_sse, misc.c:776-795 */
					if (strand) revcomp_(q, len);
					if (e < errors[c]) { errors[c] = e; id[c] = j + 1; }
				}
			}
		}
		for (int c = 0; c < 4; c++)
			if (errors[c] <= a->filter_error && res[i + c].read_type == TDO_EXTRACT_SUCCESS)
				res[i + c].read_type = (id[c] << 8) | TDO_FAIL_MATCHES_ARTIFACTS;
	}
	for (; i < end; i++) {
		uint8_t* q = seqs + offs[i];
		const int len = (int)(offs[i + 1] - offs[i]);
		int hit = 0;
		for (int j = 0; j < a->n_seq && !hit; j++) {
			const uint8_t* t = a->string + a->s_index[j];
			const int n = a->s_index[j + 1] - a->s_index[j];
			if (bpm_check_error_(t, q, n, len) <= a->filter_error) { hit = j + 1; break; }
			revcomp_(q, len);
			const int e = bpm_check_error_(t, q, n, len);
			revcomp_(q, len);
			if (e <= a->filter_error) { hit = j + 1; break; }
		}
		if (hit && res[i].read_type == TDO_EXTRACT_SUCCESS) res[i].read_type = (hit << 8) | TDO_FAIL_MATCHES_ARTIFACTS;
	}
}

/* ------------------------------------------------------------------------------------------------
 * do_label_thread for one read, barcode_hmm.c:2269-2360
 * ---------------------------------------------------------------------------------------------- */
void tdo_label_read(const tdo_model* m, const tdo_params* p, tdo_workspace* ws,
                    uint8_t* seq, uint8_t* qual, int len, int8_t* labels, tdo_result* res)
{
	res->read_type = 0;  /* clear_read_info, io.c:2084-2094 */
	res->barcode = -1;
	res->fingerprint = -1;
	int woff = 0, wlen = len;
	if (p->matchend > 0 && p->matchstart >= 0 && p->matchend > p->matchstart) {
		/* do_label_thread :2290-2296.  The reference reads past the terminator of a read that ends before matchend
		 * (undefined); here -- and on the device -- such a read is decoded on what it has inside the window. */
		const int e = len < p->matchend ? len : p->matchend;
		woff = p->matchstart;
		wlen = e > woff ? e - woff : 0;
		memset(labels, 0, (size_t)len + 1);     /* ri->labels as read_fasta_fastq() leaves it, io.c:1755-1764 */
	}
	res->b_score = tdo_backward(m, ws, seq + woff, wlen);
	tdo_forward_decode(m, ws, seq + woff, wlen, res->b_score, &res->f_score, &res->r_score, &res->bar_prob, labels);
	res->Q = tdo_qvalue(res->f_score, res->r_score, res->bar_prob);
	tdo_extract_window(m, p, seq, qual, len, woff, wlen, labels, res->Q, &res->read_type, &res->barcode, &res->fingerprint);
	if (p->dust && tdo_dust(seq, len, p->dust)) res->read_type = TDO_FAIL_LOW_COMPLEXITY;
}


/* ------------------------------------------------------------------------------------------------
 * Test support for the device kernel's position pruning (tagdust_amd/csrc/td_spec_kernel.inc): the host states
 * read-independent upper bounds on DP values -- for the first n_seg segments fb[i] on M/I_forward at position i, bwb[k]
 * on M/I_backward with k bases to go, wa[i] - 15.75 on previous_silent[i-1] + silent_to_I of segment n_seg; for the
 * segments from sfx_first on fbs[i] / bws[k] likewise and wc[k] - 15.75 on P_backward_next[i+1] + ISKIP of segment
 * sfx_first - 1 with k = len - i - 1.  tab holds the eight tables (fb, bwb, wa, wb, fbs, bws, wc, wd) of `stride` floats.
 * This runs backward() and forward_max_posterior_decoding() as above on every read and returns the smallest margins
 * bound - value it meets (a negative margin is a violated bound); -inf values are skipped.
 * margins[0..2]: forward, backward, entry term (leading segments); [3..5]: forward, backward, exit term (trailing).
 * ---------------------------------------------------------------------------------------------- */
int tdo_bound_margins(const tdo_model* m, const uint8_t* seqs, const int64_t* offs, int64_t n_reads, int n_seg, int sfx_first,
                      const float* tab, int stride, double* margins)
{
	int max_len = 1;
	for (int64_t r = 0; r < n_reads; r++) if (offs[r + 1] - offs[r] > max_len) max_len = (int)(offs[r + 1] - offs[r]);
	if (max_len + 2 > stride || n_seg < 0 || n_seg >= m->S || sfx_first < 1 || sfx_first > m->S) return -1;
	const float *fb = tab, *bwb = tab + stride, *wa = tab + 2 * stride, *fbs = tab + 4 * stride, *bws = tab + 5 * stride, *wc = tab + 6 * stride;
	tdo_workspace* ws = tdo_workspace_new(m, max_len);
	if (!ws) return -1;
	int8_t* labels = (int8_t*)malloc((size_t)max_len + 2);
	for (int k = 0; k < 6; k++) margins[k] = 1.0e30;
	const int st = ws->stride;
	const int c_end = m->col_off[n_seg];                                      /* columns of the first n_seg segments */
	const int c_sfx = sfx_first < m->S ? m->col_off[sfx_first] : m->C;       /* first column of the trailing segments */
	for (int64_t r = 0; r < n_reads; r++) {
		const uint8_t* seq = seqs + offs[r];
		const int len = (int)(offs[r + 1] - offs[r]);
		if (len < 1) continue;
		float f, rs, bp;
		const float b = tdo_backward(m, ws, seq, len);
		if (!(b > NEG_INF)) continue;
		tdo_forward_decode(m, ws, seq, len, b, &f, &rs, &bp, labels);
		for (int c = 0; c < m->C; c++) {
			if (c >= c_end && c < c_sfx) continue;
			const int o = c < c_end ? 0 : 3;
			const float* F = c < c_end ? fb : fbs;
			const float* B = c < c_end ? bwb : bws;
			for (int i = 1; i <= len; i++) {
				const float vf[2] = { ws->MF[c * st + i], ws->IF[c * st + i] };
				const float vb[2] = { ws->MB[c * st + i], ws->IB[c * st + i] };
				for (int k = 0; k < 2; k++) {
					if (vf[k] > NEG_INF && (double)F[i] - vf[k] < margins[o]) margins[o] = (double)F[i] - vf[k];
					if (vb[k] > NEG_INF && (double)B[len - i] - vb[k] < margins[o + 1]) margins[o + 1] = (double)B[len - i] - vb[k];
				}
			}
		}
		if (n_seg > 0) {
			for (int i = 1; i <= len; i++) {
				const float pm = ws->SF[(n_seg - 1) * st + (i - 1)] + m->sI[c_end];
				if (pm > NEG_INF && ((double)wa[i] - 15.75) - pm < margins[2]) margins[2] = ((double)wa[i] - 15.75) - pm;
			}
		}
		if (sfx_first < m->S) {
			const float iskip = m->trans[(size_t)m->col_off[sfx_first - 1] * 9 + TDO_ISKIP];
			for (int i = 1; i < len; i++) {
				const float pn = ws->SB[sfx_first * st + (i + 1)] + iskip;
				if (pn > NEG_INF && ((double)wc[len - i - 1] - 15.75) - pn < margins[5]) margins[5] = ((double)wc[len - i - 1] - 15.75) - pn;
			}
		}
	}
	free(labels);
	tdo_workspace_free(ws);
	return 0;
}

/* How far into a read the leading segments' posteriors really reach (tools/cut_slack.py: the slack of the device kernel's
 * bound-based pruning cut): for every read the last position at which some M / I state of the first n_seg segments has
 * forward + backward - b_score above `floor_` (the reference's posterior is exactly 0.0f below -103.98), into last_pos[r]
 * (0: none); prof[i] = the largest such term at position i over all reads; prof has 3 x prof_len entries (prof_len >= max_len + 1):
 * the terms, then the largest forward value and the largest (backward - b_score) per position. */
int tdo_lead_profile(const tdo_model* m, const uint8_t* seqs, const int64_t* offs, int64_t n_reads, int n_seg, float floor_,
                     int32_t* last_pos, float* prof, int prof_len)
{
	float* const pf = prof + prof_len;          /* [prof_len .. 2 prof_len): largest forward value, then largest backward - b_score */
	float* const pb = prof + 2 * prof_len;
	for (int i = 0; i < 2 * prof_len; i++) pf[i] = NEG_INF;
	int max_len = 1;
	for (int64_t r = 0; r < n_reads; r++) if (offs[r + 1] - offs[r] > max_len) max_len = (int)(offs[r + 1] - offs[r]);
	if (n_seg < 1 || n_seg >= m->S || prof_len < max_len + 1) return -1;
	tdo_workspace* ws = tdo_workspace_new(m, max_len);
	if (!ws) return -1;
	int8_t* labels = (int8_t*)malloc((size_t)max_len + 2);
	for (int i = 0; i < prof_len; i++) prof[i] = NEG_INF;
	const int st = ws->stride;
	const int c_end = m->col_off[n_seg];
	for (int64_t r = 0; r < n_reads; r++) {
		const uint8_t* seq = seqs + offs[r];
		const int len = (int)(offs[r + 1] - offs[r]);
		last_pos[r] = 0;
		if (len < 1) continue;
		float f, rs, bp;
		const float b = tdo_backward(m, ws, seq, len);
		if (!(b > NEG_INF)) continue;
		tdo_forward_decode(m, ws, seq, len, b, &f, &rs, &bp, labels);
		for (int c = 0; c < c_end; c++)
			for (int i = 1; i <= len; i++) {
				const float tm = ws->MF[c * st + i] + ws->MB[c * st + i] - b, ti = ws->IF[c * st + i] + ws->IB[c * st + i] - b;
				const float t = tm > ti ? tm : ti;
				const float vf = ws->MF[c * st + i] > ws->IF[c * st + i] ? ws->MF[c * st + i] : ws->IF[c * st + i];
				const float vb = (ws->MB[c * st + i] > ws->IB[c * st + i] ? ws->MB[c * st + i] : ws->IB[c * st + i]) - b;
				if (vf > pf[i]) pf[i] = vf;
				if (vb > pb[i]) pb[i] = vb;
				if (t > prof[i]) prof[i] = t;
				if (t > floor_ && i > last_pos[r]) last_pos[r] = i;
			}
	}
	free(labels);
	tdo_workspace_free(ws);
	return 0;
}

/* Per state class (column g of leading segment j, M or I; the HMMs of a segment taken together): the largest forward value and the
 * largest (backward - b_score) per position over all reads -- out[((cls * 2 + which) * prof_len) + i], cls = 2 * (column index
 * within the leading segments, HMMs folded) + (0 M, 1 I), which = 0 forward, 1 backward.  Returns the number of classes, -1 on error. */
int tdo_lead_class_profile(const tdo_model* m, const uint8_t* seqs, const int64_t* offs, int64_t n_reads, int n_seg, float* out, int prof_len, int max_cls)
{
	int max_len = 1;
	for (int64_t r = 0; r < n_reads; r++) if (offs[r + 1] - offs[r] > max_len) max_len = (int)(offs[r + 1] - offs[r]);
	int ncls = 0;
	for (int j = 0; j < n_seg; j++) ncls += 2 * m->n_col[j];
	if (n_seg < 1 || n_seg >= m->S || prof_len < max_len + 1 || ncls > max_cls) return -1;
	tdo_workspace* ws = tdo_workspace_new(m, max_len);
	if (!ws) return -1;
	int8_t* labels = (int8_t*)malloc((size_t)max_len + 2);
	for (int64_t i = 0; i < (int64_t)ncls * 2 * prof_len; i++) out[i] = NEG_INF;
	const int st = ws->stride;
	for (int64_t r = 0; r < n_reads; r++) {
		const uint8_t* seq = seqs + offs[r];
		const int len = (int)(offs[r + 1] - offs[r]);
		if (len < 1) continue;
		float f, rs, bp;
		const float b = tdo_backward(m, ws, seq, len);
		if (!(b > NEG_INF)) continue;
		tdo_forward_decode(m, ws, seq, len, b, &f, &rs, &bp, labels);
		int cbase = 0;
		for (int j = 0; j < n_seg; j++) {
			for (int h = 0; h < m->n_hmm[j]; h++)
				for (int g = 0; g < m->n_col[j]; g++) {
					const int c = m->col_off[j] + h * m->n_col[j] + g;
					for (int i = 1; i <= len; i++) {
						float* o = out + (int64_t)(2 * (cbase + g)) * 2 * prof_len;
						if (ws->MF[c * st + i] > o[i]) o[i] = ws->MF[c * st + i];
						if (ws->MB[c * st + i] - b > o[prof_len + i]) o[prof_len + i] = ws->MB[c * st + i] - b;
						o += 2 * prof_len;
						if (ws->IF[c * st + i] > o[i]) o[i] = ws->IF[c * st + i];
						if (ws->IB[c * st + i] - b > o[prof_len + i]) o[prof_len + i] = ws->IB[c * st + i] - b;
					}
				}
			cbase += m->n_col[j];
		}
	}
	free(labels);
	tdo_workspace_free(ws);
	return ncls;
}

/* ------------------------------------------------------------------------------------------------
 * The same for the start of the device kernel's restarted sweeps (td_spec_kernel.inc "Restarted sweeps"): gq[j] / gf[t] are the
 * host's impulse-response tables (tagdust_amd.lib.spec_restart_info: leading segments 0..3, then the first four trailing
 * segments, `stride` floats each).  For every read, every position t and every leading segment j the kernel may start its
 * upper ends at  max_k (gq_j[k] + SB_nseg[t + 1 + k]) + log(len - t + 1) + 0.02  -- checked here against every M / I backward
 * value of that segment at t -- and the mirror image  max_k (gf_j[k] + SF_in[t - k]) + log(t) + 0.02  against the forward
 * values of the trailing segments.  margins[0] backward, [1] forward: the smallest bound - value met (1e30: nothing to check).
 * ---------------------------------------------------------------------------------------------- */
int tdo_restart_margins(const tdo_model* m, const uint8_t* seqs, const int64_t* offs, int64_t n_reads, int n_seg, int sfx_first,
                        const float* tab, int stride, double* margins)
{
	int max_len = 1;
	for (int64_t r = 0; r < n_reads; r++) if (offs[r + 1] - offs[r] > max_len) max_len = (int)(offs[r + 1] - offs[r]);
	if (max_len + 2 > stride || n_seg < 0 || n_seg >= m->S || sfx_first < 1 || sfx_first > m->S) return -1;
	tdo_workspace* ws = tdo_workspace_new(m, max_len);
	if (!ws) return -1;
	int8_t* labels = (int8_t*)malloc((size_t)max_len + 2);
	margins[0] = margins[1] = 1.0e30;
	const int st = ws->stride;
	for (int64_t r = 0; r < n_reads; r++) {
		const uint8_t* seq = seqs + offs[r];
		const int len = (int)(offs[r + 1] - offs[r]);
		if (len < 1) continue;
		float f, rs, bp;
		const float b = tdo_backward(m, ws, seq, len);
		if (!(b > NEG_INF)) continue;
		tdo_forward_decode(m, ws, seq, len, b, &f, &rs, &bp, labels);
		for (int j = 0; j < n_seg && j < 4; j++) {
			const float* gq = tab + (size_t)j * stride;
			const float* Q = ws->SB + (size_t)n_seg * st;
			for (int t = 1; t <= len; t++) {
				float hi = NEG_INF;
				for (int k = 0; t + 1 + k <= len + 1; k++) { const float v = gq[k] + Q[t + 1 + k]; if (v > hi) hi = v; }
				hi = hi + (logf((float)(len - t + 1)) + 0.02f);
				for (int c = m->col_off[j]; c < m->col_off[j] + m->n_hmm[j] * m->n_col[j]; c++) {
					const float vb[2] = { ws->MB[c * st + t], ws->IB[c * st + t] };
					for (int k = 0; k < 2; k++) if (vb[k] > NEG_INF && (double)hi - vb[k] < margins[0]) margins[0] = (double)hi - vb[k];
				}
			}
		}
		for (int j = sfx_first; j < m->S && j - sfx_first < 4; j++) {
			const float* gf = tab + (size_t)(4 + j - sfx_first) * stride;
			const float* P = ws->SF + (size_t)(sfx_first - 1) * st;
			for (int t = 1; t <= len; t++) {
				float hi = NEG_INF;
				for (int k = 1; k <= t; k++) { const float v = gf[k] + P[t - k]; if (v > hi) hi = v; }
				hi = hi + (logf((float)t) + 0.02f);
				for (int c = m->col_off[j]; c < m->col_off[j] + m->n_hmm[j] * m->n_col[j]; c++) {
					const float vf[2] = { ws->MF[c * st + t], ws->IF[c * st + t] };
					for (int k = 0; k < 2; k++) if (vf[k] > NEG_INF && (double)hi - vf[k] < margins[1]) margins[1] = (double)hi - vf[k];
				}
			}
		}
	}
	free(labels);
	tdo_workspace_free(ws);
	return 0;
}

/* ------------------------------------------------------------------------------------------------
 * run_pHMM(MODE_GET_LABEL) analogue, barcode_hmm.c:1895-2029
 * ---------------------------------------------------------------------------------------------- */
struct batch_job {
	const tdo_model* m; const tdo_params* p; const tdo_artifacts* art;
	uint8_t* seqs; const int64_t* offs; int8_t* labels; tdo_result* res;
	int64_t start, end; int max_len; int status;
};

static void* batch_worker(void* arg)
{
	struct batch_job* jb = (struct batch_job*)arg;
	tdo_workspace* ws = tdo_workspace_new(jb->m, jb->max_len);
	if (!ws) { jb->status = 1; return NULL; }
	/* do_label_thread, :2286-2357: label + extract every read of the range, then artifacts, then DUST */
	tdo_params nodust = *jb->p;
	nodust.dust = 0;
	for (int64_t i = jb->start; i < jb->end; i++) {
		const int len = (int)(jb->offs[i + 1] - jb->offs[i]);
		tdo_label_read(jb->m, &nodust, ws, jb->seqs + jb->offs[i], NULL, len, jb->labels + jb->offs[i] + i, &jb->res[i]);
	}
	if (jb->art && jb->art->n_seq > 0) tdo_match_artifacts(jb->art, jb->seqs, jb->offs, jb->res, jb->start, jb->end);
	if (jb->p->dust) {
		for (int64_t i = jb->start; i < jb->end; i++) {
			const int len = (int)(jb->offs[i + 1] - jb->offs[i]);
			if (tdo_dust(jb->seqs + jb->offs[i], len, jb->p->dust)) jb->res[i].read_type = TDO_FAIL_LOW_COMPLEXITY;
		}
	}
	tdo_workspace_free(ws);
	jb->status = 0;
	return NULL;
}

int tdo_label_batch(const tdo_model* m, const tdo_params* p, int n_threads,
                    uint8_t* seqs, const int64_t* offs, int64_t n_reads, int8_t* labels, tdo_result* res)
{
	return tdo_label_batch_art(m, p, NULL, n_threads, seqs, offs, n_reads, labels, res);
}

int tdo_label_batch_art(const tdo_model* m, const tdo_params* p, const tdo_artifacts* art, int n_threads,
                        uint8_t* seqs, const int64_t* offs, int64_t n_reads, int8_t* labels, tdo_result* res)
{
	if (!g_logsum_ready) tdo_init_logsum();
	if (n_threads < 1) n_threads = 1;
	if (m->H > 127) return 1;
	int max_len = 0;
	for (int64_t i = 0; i < n_reads; i++) {
		const int len = (int)(offs[i + 1] - offs[i]);
		if (len > max_len) max_len = len;
	}
	struct batch_job* jobs = (struct batch_job*)calloc((size_t)n_threads, sizeof(*jobs));
	pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(*th));
	/* :1911-1922: interval = numseq / T, last thread takes the remainder */
	const int64_t interval = n_reads / n_threads;
	int rc = 0;
	for (int t = 0; t < n_threads; t++) {
		jobs[t] = (struct batch_job){ m, p, art, seqs, offs, labels, res, t * interval,
		                              (t == n_threads - 1) ? n_reads : (t + 1) * interval, max_len, 0 };
		if (pthread_create(&th[t], NULL, batch_worker, &jobs[t])) { jobs[t].status = 2; }
	}
	for (int t = 0; t < n_threads; t++) {
		if (jobs[t].status != 2) pthread_join(th[t], NULL);
		if (jobs[t].status) rc = 1;
	}
	free(jobs); free(th);
	return rc;
}
