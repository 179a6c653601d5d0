set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r04b_parity.log 2>&1; echo "parity rc=$?" ; tail -3 gpurun_out/r04b_parity.log
timeout 900 tools/ab_libs.sh "--set persample --rounds 6 --iters 10" 2 tools/bin/libdoppler_hip_prev.so tools/bin/libdoppler_hip_occ8.so > gpurun_out/r04b_ab_persample.log 2>&1
tail -40 gpurun_out/r04b_ab_persample.log
ONLY='legacy|track replay 600' SUFFIX=_legacy2 timeout 900 tools/table_rocprof.sh > /dev/null 2>&1; cat gpurun_out/r04_table_rocprof_legacy2.md
timeout 300 python tools/offset_probe.py track > gpurun_out/r04b_offset_track.log 2>&1; cat gpurun_out/r04b_offset_track.log
