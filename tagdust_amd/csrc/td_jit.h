// td_jit.h -- interface between the C-ABI layer (td_api.hip) and the model-specialised kernel builder (td_jit.hip)
#pragma once
#include <string>
#include <vector>
#include "../../include/tagdust_hip.h"
#include "td_device.h"

int td_spec_group_size(int n_hmm, int n_col, int target_override);
int td_spec_pure_last(const td_model_desc* m, int j, int col_off);
int td_spec_drop_m(const td_model_desc* m, int j, int col_off);      /* 1: the column before the pure last column spills I_backward only */
int td_spec_rt_prefix(const td_model_desc* m, int j, int col_off, int8_t* base, float* hi, float* lo, float* nv);
int td_spec_block_threads(void);
int td_spec_min_waves(void);
int td_spec_first_labels(const td_model_desc* m);   /* labels whose posteriors the forward sweep sums up itself (0: none) */
int td_spec_prune_segs(const td_model_desc* m);    /* leading segments the forward sweep may cut short (0: none), td_spec_kernel.inc "Position pruning" */
int td_spec_prune_sfx(const td_model_desc* m);     /* first trailing segment the backward sweep may cut short (S: none) */
float td_spec_prune_z(const td_model_desc* m, int n_seg, int sfx_first);
#define TD_PRUNE_TABLES 16   /* 8 position-pruning tables, then up to 4 + 4 impulse-response tables of the restarted sweeps */
#define TD_PRUNE_RESTART_MAX 4   /* leading / trailing segments a restart can bridge */
int td_spec_restart(const td_model_desc* m);   /* 1: the specialised kernel restarts the far sweeps of the pruned segments (TDS_RESTART) */
void td_spec_prune_tables(const td_model_desc* m, int n_seg, int sfx_first, int lcap, int stride, std::vector<float>& tab);
int td_spec_lsum_oob(void);      /* 1: clamp-free logsum (LDS out-of-range reads as 0), see td_spec_kernel.inc */
std::string td_spec_model_section(const td_model_desc* m, int lsum_oob = -1, int window = 0);   /* lsum_oob < 0: td_spec_lsum_oob(); window: -start/-end support compiled in */
std::string td_spec_full_source(const td_model_desc* m, int lsum_oob = -1, int window = 0);
void td_spec_layout(TdSpecLayout& L, const td_model_desc* m, int lmax);
std::string td_spec_cache_dir(void);   /* on-disk cache of compiled kernels ("" = off) */
int td_spec_compile(const td_model_desc* m, std::vector<char>& code, std::string& log, int lsum_oob = -1, int window = 0);
