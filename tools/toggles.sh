cd /root/repo
for e in "" "SELAVI_WGRAD_STREAM=0" "SELAVI_OVERLAP_AUDIO=0" "SELAVI_FUSE_BNR=0" "SELAVI_WGRAD_STREAM=0 SELAVI_OVERLAP_AUDIO=0"; do
  echo "== $e"; env $e python tools/step16_bench.py 16 16 20 fp32 2>&1 | tail -1 | cut -c1-200
done
