# A/B of two library builds of K3 on ONE box (rowbowt_amd/librbg_<A>.so against librbg_<B>.so, built side by side -- e.g. in a git worktree of the commit before):
# alternating pairs on the bench index at 4-byte positions, the same forced to 8-byte positions, and the pangenome preset (r = 1.2e8, 8-byte positions).
# usage: bash tools/k3_lean_ab.sh [A = r3stage] [B = lean]   -> gpurun_out/r06lean/     (r3stage = round 3's staging, 26.6 KB of LDS per workgroup; lean = 19.3 KB;
# ring = lean + flush windows on 64-byte boundaries of the output array: profiles/r06_k3_lean_ab.txt, r06_k3_ring_ab.txt)
A=${1:-r3stage}
B=${2:-lean}
set -u
out=gpurun_out/r06lean
mkdir -p $out
keep=$(mktemp)
cp rowbowt_amd/librbg.so $keep
trap 'cp $keep rowbowt_amd/librbg.so; rm -f $keep' EXIT
one() {  # tag, library, extra bench args
  tag=$1; lib=$2; shift 2
  cp rowbowt_amd/librbg_$lib.so rowbowt_amd/librbg.so || return 1
  python bench.py --steps 10 --warmup 2 --no-pangenome-shape --no-cpu-baseline --no-space-speed --no-markers --check-reads 2000 --property-reads 0 "$@" > $out/$tag.json 2> $out/$tag.log || { echo "FAILED $tag"; tail -5 $out/$tag.log; return 1; }
  python - <<P
import json
d=json.loads(open("$out/$tag.json").read().strip().splitlines()[-1])
print("$tag", "%.4e" % d["value"], {k.split("(")[0]: round(v["ms"],3) for k,v in d["kernels"].items()}, "parity", d.get("parity",{}).get("bit_exact_vs_oracle"))
P
}
for rep in 1 2 3; do
  one p4_${A}_$rep $A || exit 1
  one p4_${B}_$rep $B || exit 1
done
for rep in 1 2; do
  one p8_${A}_$rep $A --pos-bytes 8 || exit 1
  one p8_${B}_$rep $B --pos-bytes 8 || exit 1
done
for rep in 1 2 3; do
  for lib in $A $B; do
    cp rowbowt_amd/librbg_$lib.so rowbowt_amd/librbg.so
    python tools/pangenome_stream.py --preset driver --check-reads 500 --property-reads 20000 --total-reads 30000000 --out-json $out/pg_${lib}_$rep.json > /dev/null 2> $out/pg_${lib}_$rep.log || { echo "FAILED preset $lib"; tail -5 $out/pg_${lib}_$rep.log; exit 1; }
    echo "preset $lib rep=$rep: $(grep 'one batch, per kernel' $out/pg_${lib}_$rep.log)"
  done
done
