#!/bin/bash
# usage: tools/kres.sh [csrc-dir] [filter]  -- VGPRs / spills / LDS of every kernel (hipcc -Rpass-analysis), no GPU needed
dir=${1:-/root/repo/dark-archon_amd/csrc}; filt=${2:-.}
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wno-unused-function $KRES_FLAGS -c -Rpass-analysis=kernel-resource-usage $dir/archon_hip.hip -o /tmp/kres.o 2>&1 |
  awk '/Function Name:/{name=$(NF-1)} / VGPRs:/{v=$(NF-1)} /VGPRs Spill:/{sp=$(NF-1)} /ScratchSize/{sc=$(NF-1)} /LDS Size/{print name, "vgpr="v, "spill="sp, "scratch="sc, "lds="$(NF-1)}' |
  sed 's/_ZN6archon//' | grep -E "$filt" | cut -c1-60,100-
