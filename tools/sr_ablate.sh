# timing ablations of csrc/conv_cl16_sr.hip (variants built with tools/build_variant.sh sr_aN conv_cl16_sr.hip -- -DSLV_SR_ABL=N)
for v in "" sr_a1 sr_a4; do echo "== variant $v"; if [ -n "$v" ]; then export SELAVI_HIP_LIB=$PWD/tools/proto/libselavi_$v.so; fi; python tools/conv16_bench.py l1.spatial 20 64 2>&1 | tail -2 | head -1 | sed 's/|  *0\.[0-9]* (.*wgrad tile/| wgrad tile/;' ; done
