# NOTE: round-2 recipe.  The *_NODMA / *_NOMMA / ... ablation builds of the probes it replays were removed from the product headers in
# round 3 (VERDICT hygiene item); it runs against the round-2 tree (git history), its output is profiles/r02_power_by_component.txt.
# Package power and shader clock while ONE kernel (or one of its ablation builds) is replayed back to back: what each consumer of a
# main-loop step -- LDS-DMA stream, fragment reads, MFMAs -- costs in watts.  rocm-smi sampled at 2 Hz beside three runs of the probe.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P=diffusion-based-motion-style-transfer_amd/csrc/probes/bin
for b in attn_clock attn_clock_NODMA attn_clock_NOMMA attn_clock_NOREAD attn_clock_READONLY attn_clock_NODMA_NOREAD attn_clock_NODMA_READONLY tail_clock tail_clock_NODMA tail_clock_NOMMA tail_clock_NODMA_NOREAD tail_clock_NODMA_READONLY; do
  ( for i in 1 2 3 4; do timeout -k 10 60 $P/$b > /dev/null; done ) &
  BP=$!
  sleep 2.5
  S=""; for i in 1 2 3 4 5 6 7 8; do S="$S $(rocm-smi --showpower --showclocks 2>/dev/null | grep -i 'Package Power\|sclk' | sed 's/.*(\([0-9]*\)Mhz).*/\1/; s/.*(W): //' | tr '\n' ',')"; sleep 0.5; done
  wait $BP
  echo "$b $S"
done > gpurun_out/power_by_component.txt 2>&1
python3 - <<'PY'
import re
for line in open('gpurun_out/power_by_component.txt'):
    name, *rest = line.split()
    clk=[]; pw=[]
    for tok in rest:
        parts=[p for p in tok.split(',') if p]
        if len(parts)>=2:
            clk.append(float(parts[0])); pw.append(float(parts[1]))
    if pw: print(f"{name:28s} power {sum(pw)/len(pw):7.0f} W (min {min(pw):.0f}, max {max(pw):.0f})   sclk {sum(clk)/len(clk):6.0f} MHz")
PY
