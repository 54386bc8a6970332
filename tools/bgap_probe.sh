export TD_SPEC_CACHE_DIR=/tmp/tdc; mkdir -p gpurun_out/r4f
for o in "-DTDS_PRUNE_BGAP=15.0f -DTDS_PRUNE_MARGIN=1" "-DTDS_PRUNE_BGAP=5.0f -DTDS_PRUNE_MARGIN=1" "-DTDS_PRUNE_BGAP=0.0f -DTDS_PRUNE_MARGIN=0" "-DTDS_PRUNE_BGAP=-10.0f -DTDS_PRUNE_MARGIN=0"; do
  echo "== $o"; TD_SPEC_EXTRA_OPTS="$o" timeout -k 10 200 python3 tools/cut_probe.py c3 2>&1 | grep "prune 1" | cut -c1-90,400-900
done > gpurun_out/r4f/bgap_c3.txt 2>&1
cat gpurun_out/r4f/bgap_c3.txt
