#!/bin/bash
# builds tools/probe/bin/attn_probe_<v> for the variants given (default: all); needs csrc/obj/core.o (csrc/build.sh)
cd "$(dirname "$0")/../.."
mkdir -p tools/probe/bin
for v in ${@:-0 1 2 3 4 5 6}; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -w -DATTN_PROBE=$v -c tools/probe/attn_probe.hip -o /tmp/attn_probe_$v.o &&
    hipcc --offload-arch=gfx950 /tmp/attn_probe_$v.o e4t-diffusion_amd/csrc/obj/core.o -o tools/probe/bin/attn_probe_$v ) &
done
wait
ls tools/probe/bin
