# how often does a (two-rank, bit-identity) test fail -- or need its one retry?  usage: tools/flake.sh <runs> <pytest -k expression> [ENV=VALUE ...]
# (one line per run as it ends: a call that is cut off still leaves its partial count)
n=$1; k=$2; shift 2
f=0; w=0
for i in $(seq 1 $n); do
  env "$@" timeout 600 python -m pytest tests/test_native_comm_gpu.py tests/test_cluster_gpu.py -q -s -k "$k" > /tmp/flake_o.txt 2>&1
  if grep -q " failed" /tmp/flake_o.txt; then f=$((f+1)); grep -E "^E  " /tmp/flake_o.txt | head -1 | cut -c1-700; fi
  if grep -q "WARNING: .* audio tensors differed" /tmp/flake_o.txt; then w=$((w+1)); fi
  echo "run $i: $(tail -1 /tmp/flake_o.txt)  [failures so far $f, first-attempt mismatches $w]"
done
echo "[$k] $* : $f failures, $w first-attempt mismatches of $n"
