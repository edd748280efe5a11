#!/bin/bash
# One gpurun call: the new audio-trunk parity tests, the 16-bit experiment on them, smoke(), and the default bench line.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 420 python -m pytest tests/test_audio_archs_gpu.py -x -q -s -m gpu > gpurun_out/audio_archs_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/audio_archs_tests.txt
timeout 240 python tools/audio_archs_bf16.py > gpurun_out/audio_archs_bf16.txt 2>&1
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.txt 2>&1
echo "smoke rc=$?" >> gpurun_out/smoke.txt
timeout 400 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
tail -5 gpurun_out/audio_archs_tests.txt; cat gpurun_out/audio_archs_bf16.txt | tail -12; tail -2 gpurun_out/smoke.txt; cut -c1-600 gpurun_out/bench_final.json
