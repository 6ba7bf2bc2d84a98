# dev: A/B two attention.hip variants on one box.  usage: ab_attn.sh old.hip new.hip
set -e
for v in "$1" "$2" "$1" "$2"; do
  cp "$v" kuzushiji-vision_amd/csrc/attention.hip
  make -C kuzushiji-vision_amd/csrc -j8 > /dev/null 2>&1
  echo "=== $v"; python tools/dev/attn_bench.py 2>&1 | grep "p=0.1"
done
