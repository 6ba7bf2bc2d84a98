# dev: A/B two source variants of one csrc file on one box.  usage: ab_lib.sh <file in csrc> <old copy> <new copy> -- <bench command...>
set -e
f=$1; old=$2; new=$3; shift 4
for v in "$old" "$new" "$old" "$new"; do
  cp "$v" kuzushiji-vision_amd/csrc/$f
  make -C kuzushiji-vision_amd/csrc -j8 > /dev/null 2>&1
  echo "=== $v"; "$@" 2>&1 | grep -v amdgpu.ids
done
