set -u
for T in 376 251 200 126; do echo "== T=$T"; timeout 120 tools/att_bench 64 $T 50 2>&1 | grep "^variant" | cut -c1-120; done
echo "== B=256 T=126"; timeout 120 tools/att_bench 256 126 30 2>&1 | grep "^variant" | cut -c1-120
