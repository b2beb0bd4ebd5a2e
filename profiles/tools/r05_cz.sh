# planes per brick unit on the 64 M box, one box back to back
O=gpurun_out/r05_cz; mkdir -p $O
for cz in 32 64 128 32 64; do
  HQ_BRICK_CZ=$cz python bench.py --no-cpu-baseline --no-pmc --no-parity > $O/bench_c3_cz$cz.json 2>/dev/null; echo cz $cz; cut -c150-260 $O/bench_c3_cz$cz.json
done
