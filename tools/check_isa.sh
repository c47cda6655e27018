#!/bin/bash
# Static resource report of every kernel of libe4t_hip.so (no GPU needed: hipcc cross-compiles): VGPRs / AGPRs / scratch bytes per
# lane / occupancy / LDS per workgroup from -Rpass-analysis=kernel-resource-usage, plus the two ISA pathologies found in round 3,
# (product build: gemm_ps.hip and the other E4T_EXPERIMENTAL variants are not part of it) counted per kernel from the assembly: scratch spills inside loops and "waterfall" loops around buffer_load ... lds (a scalar
# offset the compiler could not prove uniform).  usage: tools/check_isa.sh [out]   (default profiles/rNN_isa_resources.txt)
cd "$(dirname "$0")/../e4t-diffusion_amd/csrc"
OUT=${1:-../../profiles/r04_isa_resources.txt}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -w"
echo "# $(git -C ../.. rev-parse --short HEAD)  hipcc $FLAGS" > $OUT
for f in gemm attention norm wo elementwise image core; do
  EXTRA=""; [ $f = image ] && EXTRA="-ffp-contract=off"
  hipcc $FLAGS $EXTRA -Rpass-analysis=kernel-resource-usage -c $f.hip -o /tmp/isa_$f.o 2> /tmp/isa_$f.rpt &
  hipcc $FLAGS $EXTRA -S --cuda-device-only $f.hip -o /tmp/isa_$f.s 2> /dev/null &
done
wait
python3 - "$OUT" <<'PY'
import re, subprocess, sys
out = open(sys.argv[1], "a")
out.write("file,kernel,vgprs,agprs,scratch_bytes_per_lane,occupancy_waves_per_simd,lds_bytes_per_block,waterfall_loops_around_lds_dma\n")
for f in "gemm attention norm wo elementwise image core".split():
    rpt = open(f"/tmp/isa_{f}.rpt").read()
    asm = open(f"/tmp/isa_{f}.s").read().split("\n")
    # waterfall loops per kernel symbol
    wf, cur = {}, None
    for i, line in enumerate(asm):
        m = re.match(r"^(_Z\w+):", line)
        if m: cur = m.group(1); wf.setdefault(cur, 0)
        if cur and "buffer_load_dwordx4" in line and " lds" in line and any("s_and_saveexec" in l for l in asm[max(0, i - 3):i]):
            wf[cur] += 1
    for blk in rpt.split("Function Name: ")[1:]:
        name = blk.split()[0]
        g = lambda k: (re.search(k + r"[^:]*: (\d+)", blk) or [None, "?"])[1]
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        out.write(f"{f}.hip,\"{dem}\",{g('VGPRs')},{g('AGPRs')},{g('ScratchSize')},{g('Occupancy')},{g('LDS Size')},{wf.get(name, 0)}\n")
out.close()
rows = open(sys.argv[1]).read().splitlines()[2:]
bad = [r for r in rows if int(r.rsplit(",", 1)[1]) > 0]
spill = [r for r in rows if r.split(",")[-4] not in ("0", "?")]
print(f"{len(rows)} kernels; {len(spill)} with scratch; {len(bad)} with waterfall loops around LDS-DMA")
for r in spill: print("  scratch:", r)
for r in bad: print("  waterfall:", r)
PY
