// td_model.cpp -- host-side model construction (declared in include/tagdust_model.h): own restatement of how the
// reference turns a read architecture + sequence statistics into the tables of struct model_bag.  Pure host code.
//
// The float/double mixing below is deliberate and follows the reference expression by expression: prob2scaledprob()
// takes a *float* parameter (src/misc.c:85), so every double-valued argument expression is narrowed to float before the
// (double) log(); sequencer_error_rate / indel_frequency are floats (src/interface.h:119-120) that become doubles only
// when passed to set_hmm_transition_parameters().
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "../../include/tagdust_model.h"
#include "td_jit.h"

namespace {

const double INV_SQRT_2PI = 0.3989422804014327; // src/misc.h:75

float p2sp(float p) { return (p == 0.0) ? -INFINITY : (float)log((double)p); }   // prob2scaledprob, misc.c:85-92
float sp2p(float p) { return (p == -INFINITY) ? 0.0f : (float)exp((double)p); }  // scaledprob2prob, misc.c:98-105

float logsum_f(float a, float b) // logsum, misc.c:72-78
{
	const float* T = td_logsum_table();
	const float mx = (a > b) ? a : b;
	const float mn = (a < b) ? a : b;
	if (mn == -INFINITY || (mx - mn) >= 15.7f) return mx;
	return mx + T[(int)((mx - mn) * 1000.0f)];
}

double gaussian_pdf(double x, double m, double s) // misc.c:375-379
{
	const double a = (x - m) / s;
	return INV_SQRT_2PI / s * exp(-0.5 * a * a);
}

int nuc_code(char ch) // init_nuc_code, nuc_code.c:46-74
{
	switch (ch) {
	case 'A': case 'a': return 0;
	case 'C': case 'c': return 1;
	case 'G': case 'g': return 2;
	case 'T': case 't': case 'U': case 'u': return 3;
	case '.': return 5;
	default: return 4;
	}
}

struct Col { float t[9]; float eM[5]; float eI[5]; };
struct Seg {
	int n_hmm = 0, n_col = 0;
	float skip = -INFINITY;
	std::vector<Col> cols;      // [n_hmm * n_col]
	std::vector<float> sM, sI;  // [n_hmm * n_col]
	Col& col(int f, int g) { return cols[(size_t)f * n_col + g]; }
};

enum { MM = 0, MI, MD, II, IM, DD, DM, MSKIP, ISKIP };

// set_hmm_transition_parameters(), barcode_hmm.c:1710-1881, for HMM f of a segment
void set_transitions(Seg& sg, int f, int len, double base_error, double indel_freq, double mean, double stdev)
{
	double sum_prob = 0.0;
	if (mean > 0.0 && stdev > 0.0)
		for (int i = 0; i <= len; i++) sum_prob += gaussian_pdf(i, mean, stdev);
	auto mskip = [&](double at) -> float {
		if (mean == -1.0 && stdev == -1.0) return p2sp(0.0);
		if (mean > -1.0 && stdev == -1.0) return p2sp(mean / (float)(len - 1));
		return p2sp(gaussian_pdf(at, mean, stdev) / sum_prob);
	};
	auto dead = [&](Col& c) {
		c.t[MM] = p2sp(0.0f); c.t[MI] = p2sp(0.0f); c.t[MD] = p2sp(0.0f); c.t[MSKIP] = p2sp(1.0);
		c.t[II] = p2sp(0.00); c.t[IM] = p2sp(0.0); c.t[ISKIP] = p2sp(0.0f);
		c.t[DD] = p2sp(0.0f); c.t[DM] = p2sp(0.0f);
	};
	if (len == 1) {
		dead(sg.col(f, 0));
		return;
	}
	// columns 0 .. len-2 share one shape; what differs is the MI/MD split, DD/DM and the position fed to the Gaussian
	auto live = [&](Col& c, double at, double mi_scale, double md_scale, float dd, float dm) {
		c.t[MSKIP] = mskip(at);
		const float x = p2sp(1.0 - sp2p(c.t[MSKIP]));
		c.t[MM] = p2sp(1.0 - base_error * indel_freq) + x;
		c.t[MI] = p2sp(base_error * indel_freq * mi_scale) + x;
		c.t[MD] = p2sp(base_error * indel_freq * md_scale) + x;
		c.t[II] = p2sp(1.0 - 0.999);
		c.t[IM] = p2sp(0.999);
		c.t[ISKIP] = p2sp(0.0f);
		c.t[DD] = dd;
		c.t[DM] = dm;
	};
	if (len == 2) {
		// :1744-1783 -- "base_error * indel_freq" without a 0.5 split, MD = log(0) + X
		Col& c = sg.col(f, 0);
		c.t[MSKIP] = mskip(0);
		const float x = p2sp(1.0 - sp2p(c.t[MSKIP]));
		c.t[MM] = p2sp(1.0 - base_error * indel_freq) + x;
		c.t[MI] = p2sp(base_error * indel_freq) + x;
		c.t[MD] = p2sp(base_error * indel_freq * 0.0) + x;
		c.t[II] = p2sp(1.0 - 0.999); c.t[IM] = p2sp(0.999); c.t[ISKIP] = p2sp(0.0f);
		c.t[DD] = p2sp(0.0); c.t[DM] = p2sp(0.0);
		dead(sg.col(f, 1));
		return;
	}
	live(sg.col(f, 0), 0, 0.5, 0.5, p2sp(0.0), p2sp(0.0));                                        // :1785-1808
	for (int i = 1; i < len - 2; i++) live(sg.col(f, i), i, 0.5, 0.5, p2sp(1.0 - 0.999), p2sp(0.999)); // :1811-1836
	live(sg.col(f, len - 2), len - 1.0, 1.0, 0.0, p2sp(0.0), p2sp(1.0));                          // :1839-1862
	dead(sg.col(f, len - 1));                                                                      // :1866-1878
}

// init_model_according_to_read_structure(), barcode_hmm.c:4689-5084
void init_segment(Seg& sg, const td_arch* a, int key, float base_error, float indel_freq, const double* background, int assumed_length)
{
	const int n = a->n_seq[key], len = a->seq_len[key];
	sg.n_hmm = n; sg.n_col = len;
	sg.cols.assign((size_t)n * len, Col());
	sg.sM.assign((size_t)n * len, p2sp(0.0f));
	sg.sI.assign((size_t)n * len, p2sp(0.0f));
	sg.skip = p2sp(0.0f);
	for (int i = 0; i < n; i++) {
		const char* tmpl = a->seqs[key][i];
		for (int j = 0; j < len; j++) {
			Col& c = sg.col(i, j);
			int cur = nuc_code(tmpl[j]);
			if (cur < 4) {
				for (int x = 0; x < 4; x++) {
					c.eM[x] = (x == cur) ? p2sp(1.0 - sp2p(background[4]) - base_error * (1.0 - indel_freq))
					                     : p2sp(base_error * (1.0 - indel_freq) / 3.0);
					c.eI[x] = background[x];
				}
				c.eM[4] = background[4];
				c.eI[4] = background[4];
			} else if (cur == 4) {
				for (int x = 0; x < 5; x++) { c.eM[x] = background[x]; c.eI[x] = background[x]; }
			} else { // '.'
				for (int x = 0; x < 5; x++) { c.eM[x] = (x == 4) ? p2sp(1.0) : p2sp(0.0); c.eI[x] = background[x]; }
			}
		}
		set_transitions(sg, i, len, base_error, indel_freq, -1.0, -1.0);
	}
	const char type = a->type[key];
	if (type == 'B' || type == 'F' || type == 'S') { // :4897-4921
		for (int i = 0; i < n; i++) sg.sM[(size_t)i * len] = p2sp(1.0 / (float)n);
	}
	if (type == 'P') { // :4923-4942
		for (int i = 0; i < n; i++) {
			sg.sM[(size_t)i * len] = p2sp(1.0 / (float)n) + p2sp(1.0 - 0.01);
			for (int j = 0; j < len; j++) {
				Col& c = sg.col(i, j);
				c.t[MM] = p2sp(1.0 - base_error * indel_freq) + p2sp(0.99f);
				c.t[MI] = p2sp(base_error * indel_freq) + p2sp(0.5) + p2sp(0.99f);
				c.t[MD] = p2sp(base_error * indel_freq) + p2sp(0.5) + p2sp(0.99f);
				c.t[MSKIP] = p2sp(0.01f);
				c.t[II] = p2sp(1.0 - 0.999) + p2sp(0.99f);
				c.t[IM] = p2sp(0.999) + p2sp(0.99f);
				c.t[ISKIP] = p2sp(0.01f);
			}
		}
		sg.skip = p2sp(0.01);
	}
	if (type == 'O' || type == 'G') { // :4946-5040: insert-only segments
		for (int i = 0; i < n; i++) {
			sg.sI[(size_t)i * len] = (type == 'O') ? p2sp(1.0 / (float)n) + p2sp(0.5) : p2sp(0.8935878);
			for (int j = 0; j < len; j++) {
				Col& c = sg.col(i, j);
				for (int x = 0; x < 5; x++) { c.eI[x] = c.eM[x]; c.eM[x] = p2sp(0.0); }
			}
		}
		sg.skip = (type == 'O') ? p2sp(0.5) : p2sp(1.0 - 0.8935878);
		Col& c = sg.col(0, 0);
		c.t[MM] = p2sp(0.0); c.t[MI] = p2sp(0.0); c.t[MD] = p2sp(0.0);
		c.t[IM] = p2sp(0.0); c.t[DD] = p2sp(0.0); c.t[DM] = p2sp(0.0);
		if (type == 'O') {
			c.t[MSKIP] = p2sp(0.0);
			c.t[II] = p2sp(1.0 - 1.0 / (float)(len + 1));
			c.t[ISKIP] = p2sp(1.0 / (float)(len + 1));
		} else {
			c.t[II] = p2sp(0.195); // MSKIP / ISKIP keep their default values (:5018-5028)
		}
	}
	if (type == 'R') { // :5042-5082
		for (int i = 0; i < n; i++) sg.sI[(size_t)i * len] = p2sp(1.0 / (float)n);
		Col& c = sg.col(0, 0);
		for (int x = 0; x < 5; x++) { c.eM[x] = background[x]; c.eI[x] = background[x]; }
		c.t[MM] = p2sp(0.0); c.t[MI] = p2sp(0.0); c.t[MD] = p2sp(0.0); c.t[MSKIP] = p2sp(0.0);
		c.t[II] = p2sp(1.0 - 1.0 / (float)assumed_length);
		c.t[IM] = p2sp(0.0);
		c.t[ISKIP] = p2sp(1.0 / (float)assumed_length);
		c.t[DD] = p2sp(0.0); c.t[DM] = p2sp(0.0);
		sg.skip = p2sp(0.0);
	}
}

} // namespace

// ---------------------------------------------------------------------------------------------------------
// assign_segment_sequences(), interface.c:489-598
// ---------------------------------------------------------------------------------------------------------
extern "C" int td_arch_parse(const char* const* segments, int32_t n_segments, td_arch** out)
{
	if (!segments || !out || n_segments < 1 || n_segments > TD_MAX_SEGMENTS) return TD_FAIL;
	td_arch* a = (td_arch*)calloc(1, sizeof(td_arch));
	if (!a) return TD_FAIL;
	a->n_segments = n_segments;
	for (int j = 0; j < n_segments; j++) {
		const char* s = segments[j];
		if (!s || !strchr("RGOPSFB", s[0]) || !s[0] || s[1] != ':') { td_arch_free(a); return TD_FAIL; }
		std::vector<std::vector<char>> seqs(1);
		if (s[0] == 'R') {
			seqs[0] = { 'N' };
		} else {
			for (const char* p = s + 2; *p; p++) {
				if (*p == ',') seqs.emplace_back();
				else seqs.back().push_back(*p);
			}
		}
		if (s[0] == 'B' || s[0] == 'S') seqs.emplace_back(seqs[0].size(), 'N'); // the all-N decoy, :563-581
		a->type[j] = s[0];
		a->n_seq[j] = (int32_t)seqs.size();
		a->seq_len[j] = (int32_t)seqs[0].size();
		if (a->seq_len[j] < 1) { td_arch_free(a); return TD_FAIL; }
		a->seqs[j] = (char**)calloc(seqs.size(), sizeof(char*));
		for (size_t f = 0; f < seqs.size(); f++) {
			// all HMMs of a segment have hmms[0]'s column count (barcode_hmm.c:5822); shorter ones are padded with N
			a->seqs[j][f] = (char*)calloc((size_t)a->seq_len[j] + 1, 1);
			for (int g = 0; g < a->seq_len[j]; g++) a->seqs[j][f][g] = g < (int)seqs[f].size() ? seqs[f][g] : 'N';
		}
	}
	*out = a;
	return TD_OK;
}

extern "C" void td_arch_free(td_arch* a)
{
	if (!a) return;
	for (int j = 0; j < a->n_segments; j++) {
		if (!a->seqs[j]) continue;
		for (int f = 0; f < a->n_seq[j]; f++) free(a->seqs[j][f]);
		free(a->seqs[j]);
	}
	free(a);
}

// ---------------------------------------------------------------------------------------------------------
// get_sequence_stats(), io.c:52-300
// ---------------------------------------------------------------------------------------------------------
static int sequence_stats_limit(const td_arch* a, const uint8_t* codes, const int64_t* offs, int64_t n_reads, td_seq_stats* ssi, int64_t scan_limit);

extern "C" int td_sequence_stats(const td_arch* a, const uint8_t* codes, const int64_t* offs, int64_t n_reads, td_seq_stats* ssi)
{
	// the reference reads batches of num_query = 1 000 001 reads and stops once more than 1 000 000 were seen (io.c:184)
	return sequence_stats_limit(a, codes, offs, n_reads, ssi, 1000001);
}

// with -start / -end the average length is the window's (io.c:258-260); everything else is taken over the whole reads
extern "C" int td_sequence_stats_window(const td_arch* a, const uint8_t* codes, const int64_t* offs, int64_t n_reads,
                                        int32_t matchstart, int32_t matchend, td_seq_stats* ssi)
{
	if (td_sequence_stats(a, codes, offs, n_reads, ssi) != TD_OK) return TD_FAIL;
	if (matchstart != -1 || matchend != -1) {
		// (matchend - matchstart) * total_read / total_read, rounded like every average (io.c:259-261)
		ssi->average_length = (int)floor((double)(matchend - matchstart) + 0.5);
	}
	return TD_OK;
}

static int sequence_stats_limit(const td_arch* a, const uint8_t* codes, const int64_t* offs, int64_t n_reads, td_seq_stats* ssi, int64_t scan_limit)
{
	if (!a || !codes || !offs || !ssi || n_reads < 0) return TD_FAIL;
	memset(ssi, 0, sizeof *ssi);
	for (int i = 0; i < 5; i++) ssi->background[i] = 1.0;
	int five_len = 0, three_len = 0;
	std::vector<int> five, three;
	const int last = a->n_segments - 1;
	if (a->type[0] == 'P') {
		five_len = a->seq_len[0];
		ssi->expected_5_len = five_len;
		for (int i = 0; i < five_len; i++) five.push_back(nuc_code(a->seqs[0][0][i]));
	}
	if (a->type[last] == 'P') {
		three_len = a->seq_len[last];
		ssi->expected_3_len = three_len;
		for (int i = 0; i < three_len; i++) three.push_back(nuc_code(a->seqs[last][0][i]));
	}
	double five_s0 = 0, five_s1 = 0, five_s2 = 0, three_s0 = 0, three_s1 = 0, three_s2 = 0;
	const int64_t total_read = n_reads < scan_limit ? n_reads : scan_limit;
	for (int64_t r = 0; r < total_read; r++) {
		const uint8_t* seq = codes + offs[r];
		const int len = (int)(offs[r + 1] - offs[r]);
		// seq[len] is the loader's 0 terminator (io.c:1759); anything further out is outside the reference's domain
		auto at = [&](int k) -> int { return (k >= 0 && k < len) ? seq[k] : (k == len ? 0 : -1); };
		if (len > ssi->max_seq_len) ssi->max_seq_len = len;
		ssi->average_length += len;
		for (int j = 0; j < len; j++) ssi->background[seq[j] > 4 ? 4 : seq[j]] += 1.0f;
		if (five_len) { // longest exact match of a linker suffix against the read start, > 3 nt (:141-156)
			for (int j = 0; j <= five_len; j++) {
				int c;
				for (c = 0; c < five_len - j; c++)
					if (at(c) != five[j + c]) break;
				if (c == five_len - j && c > 3) {
					five_s0++; five_s1 += five_len - j; five_s2 += (five_len - j) * (five_len - j);
					break;
				}
			}
		}
		if (three_len) { // longest exact match of a linker prefix against the read end (:158-173)
			for (int j = 0; j <= three_len; j++) {
				int c;
				for (c = 0; c < three_len - j; c++)
					if (at(len - (three_len - j - c)) != three[c]) break;
				if (c == three_len - j && c > 3) {
					three_s0++; three_s1 += three_len - j; three_s2 += (three_len - j) * (three_len - j);
					break;
				}
			}
		}
	}
	if (five_len) {
		if (five_s0 <= 1) { ssi->mean_5_len = ssi->expected_5_len; ssi->stdev_5_len = 1.0; }
		else {
			ssi->mean_5_len = five_s1 / five_s0;
			ssi->stdev_5_len = sqrt((five_s0 * five_s2 - pow(five_s1, 2.0)) / (five_s0 * (five_s0 - 1.0)));
			if (!ssi->stdev_5_len) ssi->stdev_5_len = 10000.0;
		}
	} else { ssi->mean_5_len = -1.0; ssi->stdev_5_len = -1.0; }
	if (three_len) {
		if (three_s0 <= 1) { ssi->mean_3_len = ssi->expected_3_len; ssi->stdev_3_len = 1.0; }
		else {
			ssi->mean_3_len = three_s1 / three_s0;
			ssi->stdev_3_len = sqrt((three_s0 * three_s2 - pow(three_s1, 2.0)) / (three_s0 * (three_s0 - 1.0)));
			if (!ssi->stdev_3_len) ssi->stdev_3_len = 10000.0;
		}
	} else { ssi->mean_3_len = -1.0; ssi->stdev_3_len = -1.0; }
	ssi->average_length = (int)floor((double)ssi->average_length / (double)total_read + 0.5); // :261
	double sum = 0.0;
	for (int i = 0; i < 5; i++) sum += ssi->background[i];
	for (int i = 0; i < 5; i++) ssi->background[i] = p2sp(ssi->background[i] / sum); // :268-270 (float-valued)
	return TD_OK;
}

// ---------------------------------------------------------------------------------------------------------
// init_model_bag(), barcode_hmm.c:5760-6011
// ---------------------------------------------------------------------------------------------------------
extern "C" int td_model_build(const td_arch* a, const td_seq_stats* ssi, float e, float d, td_model_tables** out)
{
	if (!a || !ssi || !out) return TD_FAIL;
	const int S = a->n_segments;
	// expected read length, :5788-5810
	int read_length = (int)ssi->average_length;
	for (int i = 0; i < S; i++) {
		if (a->type[i] == 'G') read_length = read_length - 2;
		else if (a->type[i] == 'R') { }
		else if (a->type[i] == 'P') read_length = read_length - a->seq_len[i] / 2;
		else read_length = read_length - a->seq_len[i];
	}
	if (read_length < 20) read_length = 20;

	std::vector<Seg> seg(S);
	int H = 0, C = 0;
	for (int i = 0; i < S; i++) {
		int segment_length = 0;
		if (a->type[i] == 'G') segment_length = 2;
		if (a->type[i] == 'R') segment_length = read_length;
		init_segment(seg[i], a, i, e, d, ssi->background, segment_length);
		H += seg[i].n_hmm;
		C += seg[i].n_hmm * seg[i].n_col;
	}
	if (H > TD_MAX_HMMS) return TD_FAIL;

	if (ssi->expected_5_len) { // 5' partial segment, :5841-5904
		Seg& m = seg[0];
		double sum_prob = p2sp(0.0);
		for (int i = 0; i < m.n_hmm; i++) {
			for (int j = 0; j < ssi->expected_5_len; j++) {
				m.sM[(size_t)i * m.n_col + j] = p2sp(1.0 / (float)m.n_hmm) +
				    p2sp(gaussian_pdf(j, ssi->expected_5_len - ssi->mean_5_len, ssi->stdev_5_len));
				sum_prob = logsum_f(sum_prob, m.sM[(size_t)i * m.n_col + j]);
			}
			set_transitions(m, i, (int)ssi->expected_5_len, e, d, -1.0, -1.0);
		}
		m.skip = p2sp(gaussian_pdf(ssi->expected_5_len, ssi->mean_5_len - ssi->expected_5_len, ssi->stdev_5_len));
		sum_prob = logsum_f(sum_prob, m.skip);
		for (int i = 0; i < m.n_hmm; i++)
			for (int j = 0; j < ssi->expected_5_len; j++)
				m.sM[(size_t)i * m.n_col + j] = m.sM[(size_t)i * m.n_col + j] - sum_prob;
		m.skip = m.skip - sum_prob;
	}
	if (ssi->expected_3_len) { // 3' partial segment, :5907-5920
		double sum_prob = 0;
		for (int i = 0; i < ssi->expected_3_len; i++) sum_prob += gaussian_pdf(i, ssi->mean_3_len, ssi->stdev_3_len);
		Seg& m = seg[S - 1];
		m.skip = p2sp(gaussian_pdf(0, ssi->mean_3_len, ssi->stdev_3_len) / sum_prob);
		for (int i = 0; i < m.n_hmm; i++) {
			m.sM[(size_t)i * m.n_col] = p2sp(1.0 / (float)m.n_hmm) + p2sp(1.0 - gaussian_pdf(0, ssi->mean_3_len, ssi->stdev_3_len) / sum_prob);
			set_transitions(m, i, (int)ssi->expected_3_len, e, d, ssi->mean_3_len, ssi->stdev_3_len);
		}
	}
	for (int c = 1; c < S - 1; c++) { // internal partial segments, :5922-5932
		if (a->type[c] == 'P')
			for (int i = 0; i < seg[c].n_hmm; i++) set_transitions(seg[c], i, seg[c].n_col, e, d, 0.1, -1.0);
	}

	// flatten
	const size_t bytes = sizeof(int32_t) * (3 * (size_t)S + H) + sizeof(float) * ((size_t)S + 21 * (size_t)C + (size_t)H * H) + S + 64;
	td_model_tables* t = (td_model_tables*)calloc(1, sizeof(td_model_tables));
	uint8_t* mem = (uint8_t*)calloc(1, bytes);
	if (!t || !mem) { free(t); free(mem); return TD_FAIL; }
	t->storage = mem;
	auto take = [&](size_t n) { void* p = mem; mem += (n + 7) & ~(size_t)7; return p; };
	int32_t* n_hmm = (int32_t*)take(4 * S); int32_t* n_col = (int32_t*)take(4 * S); int32_t* finger = (int32_t*)take(4 * S);
	float* skip = (float*)take(4 * S); int8_t* type = (int8_t*)take(S);
	float* trans = (float*)take(36 * (size_t)C); float* eM = (float*)take(20 * (size_t)C); float* eI = (float*)take(20 * (size_t)C);
	float* sM = (float*)take(4 * (size_t)C); float* sI = (float*)take(4 * (size_t)C);
	int32_t* label = (int32_t*)take(4 * (size_t)H); float* A = (float*)take(4 * (size_t)H * H);
	int c = 0, h = 0;
	for (int j = 0; j < S; j++) {
		n_hmm[j] = seg[j].n_hmm; n_col[j] = seg[j].n_col; skip[j] = seg[j].skip; type[j] = a->type[j];
		finger[j] = (a->type[j] == 'F') ? a->seq_len[j] : 0;
		for (int f = 0; f < seg[j].n_hmm; f++, h++) {
			label[h] = (f << 16) | j;                                          // :5954-5965
			if (seg[j].skip != p2sp(0.0)) label[h] |= (int32_t)0x80000000;
			for (int g = 0; g < seg[j].n_col; g++, c++) {
				const Col& q = seg[j].col(f, g);
				memcpy(trans + 9 * (size_t)c, q.t, 36); memcpy(eM + 5 * (size_t)c, q.eM, 20); memcpy(eI + 5 * (size_t)c, q.eI, 20);
				sM[c] = seg[j].sM[(size_t)f * seg[j].n_col + g]; sI[c] = seg[j].sI[(size_t)f * seg[j].n_col + g];
			}
		}
	}
	for (int i = 0; i < H; i++) { // label transition matrix, :5978-6006
		int open = 1;
		for (int j = i + 1; j < H; j++) {
			float v = 0;
			if ((label[i] & 0xFFFF) + 1 == (label[j] & 0xFFFF)) v = 1;
			if (((label[i] & 0xFFFF) < (label[j] & 0xFFFF)) && open) v = 1;
			if (!(label[j] & 0x80000000)) open = 0;
			A[(size_t)i * H + j] = v;
		}
		A[(size_t)i * H + i] = 1;
	}
	td_model_desc& m = t->desc;
	m.S = S; m.H = H; m.C = C; m.avg_len = (int32_t)ssi->average_length;
	for (int i = 0; i < 5; i++) m.bg[i] = (float)ssi->background[i];
	m.n_hmm = n_hmm; m.n_col = n_col; m.skip = skip; m.seg_type = type; m.finger_len = finger;
	m.trans = trans; m.eM = eM; m.eI = eI; m.sM = sM; m.sI = sI; m.label = label; m.A = A;
	*out = t;
	return TD_OK;
}

extern "C" void td_model_tables_free(td_model_tables* t)
{
	if (!t) return;
	free(t->storage);
	free(t);
}

// ---------------------------------------------------------------------------------------------------------
// threshold calibration, calibrateQ.c:17-235
// ---------------------------------------------------------------------------------------------------------
namespace {

// glibc's rand() is random_r() of type TYPE_3: an additive feedback generator x[i] = x[i-3] + x[i-31] (mod 2^32) over a
// 31-word state seeded by the Lehmer recurrence 16807 * x mod (2^31 - 1), first 310 outputs discarded, result = x >> 1.
// The calibration draws ~10^8 numbers one at a time; an inline copy of that recurrence avoids the locked library call
// for each.  It is only used after its first outputs have been checked against this process's own srand()/rand(); any
// other C library keeps calling rand().
struct GlibcRand {
	int32_t r[34];
	int f = 3, b = 0;   // front / rear indices into the 31-word ring
	void seed(uint32_t s)
	{
		if (s == 0) s = 1;
		r[0] = (int32_t)s;
		for (int i = 1; i < 31; i++) {
			const long hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
			long word = 16807 * lo - 2836 * hi;
			if (word < 0) word += 2147483647;
			r[i] = (int32_t)word;
		}
		f = 3; b = 0;
		for (int i = 0; i < 310; i++) (void)next();
	}
	int next()
	{
		const uint32_t v = (uint32_t)r[f] + (uint32_t)r[b];
		r[f] = (int32_t)v;
		if (++f >= 31) f = 0;
		if (++b >= 31) b = 0;
		return (int)(v >> 1);
	}
	static bool matches_libc()
	{
		static const bool ok = [] {
			for (uint32_t s : { 1u, 42u, 2654435761u }) {
				GlibcRand g; g.seed(s);
				srand(s);
				for (int i = 0; i < 1000; i++) if (g.next() != rand()) return false;
			}
			return RAND_MAX == 2147483647;
		}();
		return ok;
	}
};

struct Rng {
	int kind;            // 0 = libc rand(), 1 = the reference's RTEST LCG (misc.c:878-887)
	uint32_t next = 1;
	bool own = false;
	GlibcRand g;
	void seed(uint32_t s)
	{
		if (kind) { next = s; return; }
		own = GlibcRand::matches_libc();
		if (own) g.seed(s); else srand(s);
	}
	// a plain rand() (simulate_reads.c draws barcode numbers and lengths with rand() % n)
	int irand()
	{
		if (kind) { next = next * 1103515245u + 12345u; return (int)((unsigned)(next / 65536) % 32768); }
		return own ? g.next() : rand();
	}
	// "(float)rand()/(float)my_rand_max", barcode_hmm.c:2610,2721 (my_rand_max = RAND_MAX, or 32768 under RTEST)
	long long n_draws = 0;
	double draw()
	{
		n_draws++;
		if (kind) {
			next = next * 1103515245u + 12345u;
			return (float)(int)((unsigned)(next / 65536) % 32768) * 0x1p-15f;   // / 32768.0f, exactly
		}
		// (float)RAND_MAX is 2^31 for glibc's 2^31 - 1, so the division is an exact scaling
		if (own) return (float)g.next() * 0x1p-31f;
		return (float)rand() / (float)(unsigned)RAND_MAX;
	}
};

struct ModelView { // flattened tables with the per-segment offsets the emitters need
	const td_model_desc* m;
	std::vector<int> col_off;
	explicit ModelView(const td_model_desc* d) : m(d), col_off((size_t)d->S)
	{
		int c = 0;
		for (int j = 0; j < d->S; j++) { col_off[(size_t)j] = c; c += d->n_hmm[j] * d->n_col[j]; }
	}
	int col(int seg, int hmm, int g) const { return col_off[(size_t)seg] + hmm * m->n_col[seg] + g; }
};

// The emitters compare one uniform draw against a running sum of probabilities that restarts from log 0 at every step
// and folds the same model parameters in the same order (logsum in float, scaledprob2prob through exp in double).
// Those thresholds depend on the model only, so they are evaluated once -- with exactly those operations -- and a step
// becomes a search for the first threshold above the draw: identical decisions, no exp per comparison.
struct EmitTables {
	std::vector<std::vector<double>> silent;   // per segment: thresholds after (hmm i, column j, M) and (i, j, I), in loop order
	std::vector<double> tM, tI, tD;            // per column: 3 (MM, MI, MD), 2 (II, IM), 1 (DD)
	std::vector<double> eM, eI;                // per column: 5 emission thresholds each
	double bg[5];
	EmitTables(const ModelView& v)
	{
		const td_model_desc* m = v.m;
		silent.resize((size_t)m->S);
		for (int seg = 0; seg < m->S; seg++) {
			double sum = p2sp(0.0f);
			for (int i = 0; i < m->n_hmm[seg]; i++)
				for (int j = 0; j < m->n_col[seg]; j++) {
					sum = logsum_f(sum, m->sM[v.col(seg, i, j)]); silent[(size_t)seg].push_back(sp2p(sum));
					sum = logsum_f(sum, m->sI[v.col(seg, i, j)]); silent[(size_t)seg].push_back(sp2p(sum));
				}
		}
		const size_t C = (size_t)m->C;
		tM.resize(C * 3); tI.resize(C * 2); tD.resize(C); eM.resize(C * 5); eI.resize(C * 5);
		for (size_t c = 0; c < C; c++) {
			const float* t = m->trans + c * 9;
			double sum = p2sp(0.0f);
			sum = logsum_f(sum, t[MM]); tM[c * 3 + 0] = sp2p(sum);
			sum = logsum_f(sum, t[MI]); tM[c * 3 + 1] = sp2p(sum);
			sum = logsum_f(sum, t[MD]); tM[c * 3 + 2] = sp2p(sum);
			sum = p2sp(0.0f);
			sum = logsum_f(sum, t[II]); tI[c * 2 + 0] = sp2p(sum);
			sum = logsum_f(sum, t[IM]); tI[c * 2 + 1] = sp2p(sum);
			sum = p2sp(0.0f);
			sum = logsum_f(sum, t[DD]); tD[c] = sp2p(sum);
			sum = p2sp(0.0f);
			for (int nuc = 0; nuc < 5; nuc++) { sum = logsum_f(sum, m->eM[c * 5 + (size_t)nuc]); eM[c * 5 + (size_t)nuc] = sp2p(sum); }
			sum = p2sp(0.0f);
			for (int nuc = 0; nuc < 5; nuc++) { sum = logsum_f(sum, m->eI[c * 5 + (size_t)nuc]); eI[c * 5 + (size_t)nuc] = sp2p(sum); }
		}
		double sum = p2sp(0.0f);
		for (int nuc = 0; nuc < 5; nuc++) { sum = logsum_f(sum, m->bg[nuc]); bg[nuc] = sp2p(sum); }
	}
};

// emit_read_sequence(), barcode_hmm.c:2696-3046
void emit_read(const ModelView& v, const EmitTables& T, Rng& rng, int average_length, std::vector<uint8_t>& seq)
{
	const td_model_desc* m = v.m;
	double r = rng.draw();
	size_t current_length = 0;
	seq.clear();
	while ((int)current_length < average_length) {
		int state = 0, column = 0, hmm = 0, segment = 0; // 0 silent, 1 M, 2 I, 3 D
		while (1) {
			r = rng.draw();
			switch (state) {
			case 0: {
				// first (hmm, column, M|I) whose running sum exceeds the draw; the sums never decrease, so the first
				// threshold above r is found by bisection.  No hit leaves the state silent (and draws again), as in the
				// reference's loop.
				const std::vector<double>& th = T.silent[(size_t)segment];
				const size_t k = (size_t)(std::upper_bound(th.begin(), th.end(), r) - th.begin());
				if (k < th.size()) {
					const int len = m->n_col[segment];
					state = (k & 1) ? 2 : 1;
					hmm = (int)(k / 2) / len;
					column = (int)(k / 2) % len;
				}
				break;
			}
			case 1: {
				const double* t = &T.tM[(size_t)v.col(segment, hmm, column) * 3];
				if (r < t[0]) { state = 1; column++; break; }
				if (r < t[1]) { state = 2; break; }
				if (r < t[2]) { state = 3; column++; break; }
				state = 0; segment++; column = 0; hmm = 0; // MSKIP takes whatever is left
				break;
			}
			case 2: {
				const double* t = &T.tI[(size_t)v.col(segment, hmm, column) * 2];
				if (r < t[0]) { state = 2; break; }
				if (r < t[1]) { state = 1; column++; break; }
				state = 0; segment++; column = 0; hmm = 0; // ISKIP
				break;
			}
			case 3: {
				if (r < T.tD[(size_t)v.col(segment, hmm, column)]) { state = 3; column++; break; }
				state = 1; column++; // DM
				break;
			}
			}
			r = rng.draw();
			if (state == 1 || state == 2) {
				const double* e = &(state == 1 ? T.eM : T.eI)[(size_t)v.col(segment, hmm, column) * 5];
				// first threshold above the draw, without a data-dependent branch per candidate (the thresholds ascend)
				const int nuc = (int)!(r < e[0]) + (int)!(r < e[1]) + (int)!(r < e[2]) + (int)!(r < e[3]) + (int)!(r < e[4]);
				if (nuc < 5) {
					if (seq.size() <= current_length) seq.resize(2 * current_length + 64);
					seq[current_length++] = (uint8_t)nuc;
				}
			}
			if (segment == m->S) break;
		}
		if ((int)current_length < average_length) current_length = 0;
	}
	seq.resize(current_length);
}

// emit_random_sequence(), barcode_hmm.c:2599-2680
void emit_random(const EmitTables& T, Rng& rng, int average_length, std::vector<uint8_t>& seq)
{
	size_t current_length = 0;
	double r = rng.draw();
	seq.clear();
	const double stop = 1.0 - (1.0 / (float)average_length);
	while ((int)current_length < average_length) {
		while (1) {
			const int nuc = (int)!(r < T.bg[0]) + (int)!(r < T.bg[1]) + (int)!(r < T.bg[2]) + (int)!(r < T.bg[3]) + (int)!(r < T.bg[4]);
			if (nuc < 5) {
				if (seq.size() <= current_length) seq.resize(2 * current_length + 64);
				seq[current_length++] = (uint8_t)nuc;
			}
			r = rng.draw();
			if (r > stop) break;
		}
		if ((int)current_length < average_length) current_length = 0;
	}
	seq.resize(current_length);
}

} // namespace

extern "C" int td_calibration_emit(const td_arch* a, const td_seq_stats* ssi, float d, uint32_t seed, int32_t n_reads,
                                   int32_t rng_kind, td_calibration** out)
{
	if (!a || !ssi || !out || n_reads < 4) return TD_FAIL;
	Rng rng; rng.kind = rng_kind != 0;
	rng.seed(seed);                                    // srand(seed), calibrateQ.c:33
	const int binsize = n_reads / 4;
	td_model_tables* em = nullptr;
	if (td_model_build(a, ssi, 0.05f, d, &em) != TD_OK) return TD_FAIL; // sequencer_error_rate forced to 0.05, :65
	// reads are emitted from a model whose B / S decoy HMM has prior 0 (:70-86)
	{
		int c = 0;
		float* sM = const_cast<float*>(em->desc.sM);
		for (int j = 0; j < em->desc.S; j++) {
			const int n = em->desc.n_hmm[j], nc = em->desc.n_col[j];
			if (a->type[j] == 'B' || a->type[j] == 'S') {
				for (int f = 0; f < n - 1; f++) sM[c + f * nc] = p2sp(1.0 / (float)(n - 1));
				sM[c + (n - 1) * nc] = p2sp(0.0);
			}
			c += n * nc;
		}
	}
	td_calibration* cal = (td_calibration*)calloc(1, sizeof(td_calibration));
	if (!cal) { td_model_tables_free(em); return TD_FAIL; }
	std::vector<uint8_t> all, one;
	std::vector<int64_t> offs(1, 0);
	std::vector<uint8_t> rnd;
	const ModelView view(&em->desc);
	const EmitTables tables(view);
	const int avg = (int)ssi->average_length;
	int readnum = 0;
	const auto tc0 = std::chrono::steady_clock::now();
	for (int i = 0; i < binsize * 2; i++) {             // :88-100
		emit_read(view, tables, rng, avg, one);
		all.insert(all.end(), one.begin(), one.end());
		offs.push_back((int64_t)all.size()); rnd.push_back(0);
		readnum++;
	}
	const auto tc1 = std::chrono::steady_clock::now();
	const long long d1 = rng.n_draws;
	for (int i = 0; i < binsize + binsize; i++) {       // :102-113
		emit_random(tables, rng, avg, one);
		all.insert(all.end(), one.begin(), one.end());
		offs.push_back((int64_t)all.size()); rnd.push_back(1);
		readnum++;
		if (readnum == n_reads) break;
	}
	td_model_tables_free(em);
	if (td_model_build(a, ssi, 0.05f, d, &cal->scoring) != TD_OK) { free(cal); return TD_FAIL; } // :117-119
	if (getenv("TD_CAL_DEBUG")) {
		const auto tc2 = std::chrono::steady_clock::now();
		fprintf(stderr, "calibration: model reads %.2f s (%lld draws), random reads %.2f s (%lld draws), %d reads\n",
		        std::chrono::duration<double>(tc1 - tc0).count(), d1, std::chrono::duration<double>(tc2 - tc1).count(),
		        rng.n_draws - d1, readnum);
	}
	cal->n_reads = readnum;
	cal->codes = (uint8_t*)malloc(all.size() + 1);
	cal->offs = (int64_t*)malloc(sizeof(int64_t) * offs.size());
	cal->is_random = (uint8_t*)malloc(rnd.size() + 1);
	if (!cal->codes || !cal->offs || !cal->is_random) { td_calibration_free(cal); return TD_FAIL; }
	memcpy(cal->codes, all.data(), all.size());
	memcpy(cal->offs, offs.data(), sizeof(int64_t) * offs.size());
	memcpy(cal->is_random, rnd.data(), rnd.size());
	*out = cal;
	return TD_OK;
}

extern "C" void td_calibration_free(td_calibration* c)
{
	if (!c) return;
	free(c->codes); free(c->offs); free(c->is_random);
	td_model_tables_free(c->scoring);
	free(c);
}

extern "C" float td_calibration_select(const float* mapq, const uint8_t* is_random, int64_t n)
{
	// calibrateQ.c:146-212.  qsort() of glibc 2.35 is a merge sort for arrays of this size, i.e. stable: equal Q values
	// keep their emission order (model reads before background reads).
	std::vector<int64_t> order((size_t)n);
	for (int64_t i = 0; i < n; i++) order[(size_t)i] = i;
	std::stable_sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return mapq[x] > mapq[y]; });
	double TP = 0.0, FP = 0.0, TN = 0.0, FN = 0.0;
	for (int64_t i = 0; i < n; i++) { if (is_random[i]) TN++; else FN++; }
	float best = 0.0f, thres4 = 1000.0f;
	for (int64_t k = 0; k < n; k++) {
		const int64_t i = order[(size_t)k];
		if (is_random[i]) { FP += 1.0; TN -= 1.0; } else { TP += 1.0; FN -= 1.0; }
		const float sensitivity = TP / (TP + FN);
		const float specificity = TN / (TN + FP);
		if (sensitivity + specificity > best) { best = specificity + sensitivity; thres4 = mapq[i]; }
	}
	return thres4 < 20 ? thres4 : 20.0f;
}

extern "C" int td_estimate_threshold(td_ctx* ctx, const td_arch* a, const td_seq_stats* ssi, float d, uint32_t seed,
                                     int32_t n_reads, int32_t rng, float* threshold)
{
	if (!ctx || !threshold) return TD_FAIL;
	// The scoring model's kernel is compiled (seconds) while the host emits the reads (seconds, bound to one thread by the
	// rand() sequence): the compile only fills the code cache td_model_upload looks into.
	std::thread warm;
	td_model_tables* scoring = nullptr;
	int32_t specialize = 0;
	const bool overlap = getenv("TD_CAL_OVERLAP") ? atoi(getenv("TD_CAL_OVERLAP")) != 0 : true;
	if (overlap && td_get_option(ctx, "specialize", &specialize) == TD_OK && specialize &&
	    td_model_build(a, ssi, 0.05f, d, &scoring) == TD_OK) {
		warm = std::thread([scoring] { std::vector<char> code; std::string log; (void)td_spec_compile(&scoring->desc, code, log); });
	}
	td_calibration* cal = nullptr;
	const int emitted = td_calibration_emit(a, ssi, d, seed, n_reads, rng, &cal);
	if (warm.joinable()) warm.join();
	td_model_tables_free(scoring);
	if (emitted != TD_OK) return TD_FAIL;
	int rc = TD_FAIL;
	// The simulated reads come out of the model itself: most are as long as the data, the read segment's geometric length gives
	// a few of ten times that.  The decode workspace is laid out for a batch's longest read, so the reads are scored in three
	// length classes (up to the 75th / 95th percentile / the rest): 30 GB of workspace instead of 260 GB for 400 000 reads of a
	// 150-nt architecture -- which would also leave no room for the second workspace of the pipelined calls afterwards.
	const int64_t n = cal->n_reads;
	std::vector<float> q((size_t)n);
	bool ok = td_model_upload(ctx, &cal->scoring->desc) == TD_OK;
	if (ok && n > 0) {
		std::vector<int32_t> lens((size_t)n), sorted;
		for (int64_t i = 0; i < n; i++) lens[(size_t)i] = (int32_t)(cal->offs[i + 1] - cal->offs[i]);
		sorted = lens;
		std::sort(sorted.begin(), sorted.end());
		const int32_t cut[3] = { sorted[(size_t)((n - 1) * 3 / 4)], sorted[(size_t)((n - 1) * 19 / 20)], sorted[(size_t)(n - 1)] };
		int32_t lo = -1;
		for (int k = 0; k < 3 && ok; k++) {
			if (cut[k] <= lo) continue;
			std::vector<int64_t> idx;
			std::vector<int64_t> offs(1, 0);
			std::vector<uint8_t> codes;
			for (int64_t i = 0; i < n; i++) {
				if (lens[(size_t)i] <= lo || lens[(size_t)i] > cut[k]) continue;
				idx.push_back(i);
				codes.insert(codes.end(), cal->codes + cal->offs[i], cal->codes + cal->offs[i + 1]);
				offs.push_back((int64_t)codes.size());
			}
			lo = cut[k];
			if (idx.empty()) continue;
			if (codes.empty()) codes.push_back(0);
			std::vector<td_read_result> res(idx.size());
			ok = td_batch_upload(ctx, codes.data(), offs.data(), (int64_t)idx.size()) == TD_OK && td_run(ctx, TD_MODE_GET_PROB) == TD_OK &&
			     td_batch_download(ctx, res.data(), nullptr, nullptr) == TD_OK;
			for (size_t j = 0; ok && j < idx.size(); j++) q[(size_t)idx[j]] = res[j].mapq;
		}
	}
	if (ok) {
		*threshold = td_calibration_select(q.data(), cal->is_random, n);
		rc = TD_OK;
	}
	td_calibration_free(cal);
	return rc;
}

// ---------------------------------------------------------------------------------------------------------
// simulate_reads.c:28-470 (the reference's simreads): the bench / test inputs of the architectures it can emit
// ---------------------------------------------------------------------------------------------------------
namespace {

char sim_base(Rng& rng)
{
	const double r = rng.draw();                              // simulate_reads.c:176-186
	return r < 0.25 ? 'A' : r < 0.5 ? 'C' : r < 0.75 ? 'G' : 'T';
}

// mutate(), simulate_reads.c:480-560
std::string sim_mutate(const td_sim_params* p, const std::string& seq, Rng& rng)
{
	std::string out;
	const int len = (int)seq.size();
	for (int j = 0; j < len; j++) {
		double r = rng.draw();
		if (r <= (double)p->error_rate) {
			r = rng.draw();
			if (r <= (double)p->indel_frac) {
				r = rng.draw();
				const double cutoff = (j == len - 1) ? 0.0 : 0.5;
				if (r <= cutoff) { const char n = sim_base(rng); out += seq[(size_t)j]; out += n; }   // insertion; else deletion
			} else {
				char n = seq[(size_t)j];
				while (n == seq[(size_t)j]) n = sim_base(rng);
				out += n;
			}
		} else {
			out += seq[(size_t)j];
		}
	}
	return out;
}

} // namespace

extern "C" int td_simreads(const td_sim_params* p, const char* const* barcodes, int32_t n_barcodes, char** fastq_out, int64_t* len_out)
{
	if (!p || !fastq_out || !len_out || p->numseq < 0 || p->barnum < 0 || (p->barnum > 0 && (!barcodes || n_barcodes < p->barnum))) return TD_FAIL;
	if (p->readlen < 0 || p->readlen_mod < 0 || p->end_loss < 0) return TD_FAIL;
	Rng rng;
	rng.kind = p->rng;
	rng.seed(p->seed);                                        // srand(seed), :37
	std::string out;
	out.reserve((size_t)p->numseq * (size_t)(2 * (p->readlen + 40) + 64));
	char num[48];
	const std::string seq5 = p->seq5 ? p->seq5 : "", seq3 = p->seq3 ? p->seq3 : "";
	const int n_real = (int)((double)(float)p->numseq * (1.0 - (double)p->random_frac));   // :141
	for (int i = 0; i < n_real; i++) {
		std::string tmp = seq5;
		int used = 0;
		if (p->barnum) { used = rng.irand() % p->barnum; tmp += barcodes[used]; }              // :155-159
		std::string sequenced = sim_mutate(p, tmp, rng);
		int c = p->readlen;
		if (p->readlen_mod) c = p->readlen - p->readlen_mod + rng.irand() % (p->readlen_mod * 2);   // :169-173
		std::string read;
		for (int j = 0; j < c; j++) read += sim_base(rng);
		sequenced += read;
		if (p->seq3) sequenced += sim_mutate(p, seq3, rng);                                      // :193-199
		if (p->end_loss) {                                                                       // :202-219
			int start = rng.irand() % (p->end_loss * 2);
			sequenced = start < (int)sequenced.size() ? sequenced.substr((size_t)start) : std::string();
			start = rng.irand() % (p->end_loss * 2);
			if (start > 0) sequenced = start < (int)sequenced.size() ? sequenced.substr(0, sequenced.size() - (size_t)start) : std::string();
		}
		snprintf(num, sizeof num, "@READ%d;SEQ:", i);
		out += num; out += read;
		if (p->barnum) { out += ";RBC:"; out += barcodes[used]; snprintf(num, sizeof num, ";BARNUM:%d", used + 1); out += num; }
		else out += ";BARNUM:1";
		out += '\n'; out += sequenced; out += "\n+\n"; out.append(sequenced.size(), 'I'); out += '\n';
	}
	// fully random sequences (:259-321): length = linkers + sim_barlen + read length; the end-loss draws are made (and act
	// on another buffer there), so they are consumed here too
	const int c = (int)seq5.size() + (int)seq3.size() + p->barlen + p->readlen;
	for (int i = n_real; i < p->numseq; i++) {
		std::string s;
		for (int j = 0; j < c; j++) s += sim_base(rng);
		if (p->end_loss) { (void)rng.irand(); (void)rng.irand(); }
		snprintf(num, sizeof num, "@RAND%d;SEQ:NONE;", i);
		out += num; out += p->barnum ? "RBC:NONE;BARNUM:0" : "BARNUM:0";
		out += '\n'; out += s; out += "\n+\n"; out.append((size_t)c, 'I'); out += '\n';
	}
	char* buf = (char*)malloc(out.size() + 1);
	if (!buf) return TD_FAIL;
	memcpy(buf, out.data(), out.size());
	buf[out.size()] = 0;
	*fastq_out = buf; *len_out = (int64_t)out.size();
	return TD_OK;
}

extern "C" void td_text_free(char* text) { free(text); }

// ---------------------------------------------------------------------------------------------------------
// architecture selection, test_architectures.c:20-289
// ---------------------------------------------------------------------------------------------------------
extern "C" int td_compare_architectures(td_ctx* ctx, const td_arch* const* archs, int32_t n_arch, const uint8_t* codes,
                                        const int64_t* offs, int64_t n_reads, float e, float d, int32_t n_threads,
                                        float* posterior, int32_t* best)
{
	if (!ctx || !archs || n_arch < 1 || !codes || !offs || !posterior || !best || n_reads < 1) return TD_FAIL;
	if (n_threads < 1) n_threads = 1;
	// test_architectures() sets num_query = 100 000 (:38-42): statistics then scan batches of 100 000 reads until more than
	// 1 000 000 were seen (1 100 000 reads), and the candidates are scored on the first batch only (:182-184)
	const int64_t n_score = n_reads < 100000 ? n_reads : 100000;
	// every candidate's own sequence statistics and model (test_architectures.c:60-170), then all of them over the first
	// batch in one launch of the generic kernel (td_arch_scores: no per-candidate compile, reads staged once)
	std::vector<td_model_tables*> tabs((size_t)n_arch, nullptr);
	std::vector<const td_model_desc*> descs((size_t)n_arch, nullptr);
	int rc = TD_OK;
	for (int k = 0; k < n_arch && rc == TD_OK; k++) {
		td_seq_stats st;
		if (sequence_stats_limit(archs[k], codes, offs, n_reads, &st, 1100000) != TD_OK || td_model_build(archs[k], &st, e, d, &tabs[(size_t)k]) != TD_OK) rc = TD_FAIL;
		else descs[(size_t)k] = &tabs[(size_t)k]->desc;
	}
	std::vector<float> b((size_t)n_arch * (size_t)n_score);
	if (rc == TD_OK && td_arch_scores(ctx, descs.data(), n_arch, codes, offs, n_score, b.data()) != TD_OK) rc = TD_FAIL;
	for (td_model_tables* t : tabs) td_model_tables_free(t);
	for (int k = 0; k < n_arch && rc == TD_OK; k++) {
		const float* bk = b.data() + (size_t)k * (size_t)n_score;
		float total = p2sp(1.0);                                      // ab->arch_posterior[i] = prob2scaledprob(1.0), test_architectures.c:164
		const int64_t interval = n_score / n_threads;                  // barcode_hmm.c:1911
		for (int t = 0; t < n_threads; t++) {
			const int64_t lo = t * interval, hi = (t == n_threads - 1) ? n_score : (t + 1) * interval;
			float partial = p2sp(1.0);                                 // :1935
			for (int64_t i = lo; i < hi; i++) partial += bk[i];       // do_arch_comparison :2135
			total += partial;                                          // :2003
		}
		posterior[k] = total;
	}
	if (rc != TD_OK) return TD_FAIL;
	if (n_arch > 1) {
		float sum = posterior[0];                                      // barcode_hmm.c:2009-2016
		for (int k = 1; k < n_arch; k++) sum = logsum_f(sum, posterior[k]);
		for (int k = 0; k < n_arch; k++) posterior[k] = posterior[k] - sum;
		sum = p2sp(0.0f);                                              // test_architectures.c:191-206
		for (int k = 0; k < n_arch; k++) sum = logsum_f(sum, posterior[k]);
		*best = -1;
		float best_score = -1.0f;
		for (int k = 0; k < n_arch; k++) {
			posterior[k] = sp2p(posterior[k] - sum);
			if (posterior[k] > best_score) { best_score = posterior[k]; *best = k; }
		}
	} else {
		posterior[0] = 1.0f;
		*best = 0;
	}
	return TD_OK;
}
