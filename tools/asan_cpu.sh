#!/bin/bash
# Host code under AddressSanitizer + UBSan on the CPU (GPU sanitizers are not available on the pool): builds the library
# with the sanitizers on the host side only (the device side is compiled as usual and never runs here) into /tmp and runs
# the CPU test files that exercise host logic against it -- FASTQ parse / write, model construction, calibration emitters,
# simreads, kernel source generation, shard arithmetic, the streaming pipeline's reader / parser / batcher (parse-only runs).   usage: tools/asan_cpu.sh
set -e
cd "$(dirname "$0")/.."
OUT=/tmp/td_asan
mkdir -p $OUT
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
python -c "from tagdust_amd import build; build.write_embedded()"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fPIC -shared -Wno-option-ignored \
	-fsanitize=address,undefined -fno-sanitize=vptr,function -fno-omit-frame-pointer -shared-libasan \
	tagdust_amd/csrc/td_kernels.hip tagdust_amd/csrc/td_api.hip tagdust_amd/csrc/td_stage.hip tagdust_amd/csrc/td_jit.hip \
	-x hip tagdust_amd/csrc/td_model.cpp tagdust_amd/csrc/td_fastq.cpp tagdust_amd/csrc/td_stream.cpp tagdust_amd/csrc/td_multi.cpp \
	-lhiprtc -ldl -o $OUT/libtagdust_hip_asan.so
TD_LIB_PATH=$OUT/libtagdust_hip_asan.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
	python -m pytest tests/test_io.py tests/test_model_builder.py tests/test_calibration.py tests/test_simreads.py tests/test_spec_source.py \
	tests/test_shard_gloo.py tests/test_prune.py tests/test_stream.py tests/test_abi.py -x -q -m "not gpu" -p no:cacheprovider
