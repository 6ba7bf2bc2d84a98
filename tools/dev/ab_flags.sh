# dev: A/B compile-time flags of one csrc file on one box.  usage: ab_flags.sh <file.hip> "<flags A>" "<flags B>" ... -- <bench command...>
set -e
f=$1; shift
variants=()
while [ "$1" != "--" ]; do variants+=("$1"); shift; done
shift
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast"
for rep in 1 2; do
for v in "${variants[@]}"; do
  touch kuzushiji-vision_amd/csrc/$f
  make -C kuzushiji-vision_amd/csrc FLAGS="$F $v" -j8 > /dev/null 2>&1
  echo "=== flags: [$v]"; "$@" 2>&1 | grep -v amdgpu.ids
done
done
