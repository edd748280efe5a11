# timing of csrc/wgrad_cl16_acc.hip: kernel / reduce split under rocprofv3, then the ablations (variants built with
# tools/build_variant.sh wa_aN wgrad_cl16_acc.hip -- -DSLV_WA_ABL=N: 1 no MFMA, 2 no dY DMA, 3 no X loads, 4 neither, 5 no fragment reads)
export TMPDIR=/tmp
for v in ${@:-"" wa_a1 wa_a2 wa_a3 wa_a4 wa_a5}; do echo "== variant $v"; if [ -n "$v" ] && [ "$v" != base ]; then export SELAVI_HIP_LIB=$PWD/tools/proto/libselavi_$v.so; fi; python tools/conv16_bench.py l1.spatial 20 64 2>&1 | tail -2 | head -1 | sed 's/|  *0\.[0-9]* (.*wgrad tile/| wgrad tile/;' ; done
