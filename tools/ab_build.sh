#!/bin/bash
# Same-box A/B of compile-time variants: bench, rebuild the library ON THE GPU BOX with extra flags, bench, ..., base again.
# Usage (on the box, from the repo root): [BENCH_ARGS='--precision fp8'] [RUNS=2] bash tools/ab_build.sh "-DA=1" "-DA=1 -DB=2" ...
B="python bench.py --steps 40 --warmup 4 --no-cpu-baseline $BENCH_ARGS"
run() { for i in $(seq ${RUNS:-2}); do $B 2>/dev/null | tail -1 | python3 tools/ab_line.py; done
  # a variant that computes garbage can be FASTER (fewer toggling bits at the power cap): every variant must also reproduce the embeddings
  python - <<'PY'
import torch, bench, keds_amd
m = bench.random_clip(torch.device("cuda", 0))
img = torch.randn(128, 3, 224, 224, generator=torch.Generator(device="cuda").manual_seed(1001), device="cuda")
e = m.encode_image(img, normalize=True).double()
print("   embedding checksum %.9f  finite %s" % (float((e * torch.arange(1, 769, device="cuda", dtype=torch.float64)).sum()), bool(torch.isfinite(e).all())))
PY
}
# a variant whose build fails is SKIPPED (the previous library must not run under its label); the base build is restored on
# any exit; EXTRA is recorded in the library (keds_build_flags) and in the evidence digest
restore() { (cd keds_amd/csrc; make -j8 EXTRA="" > /tmp/mk.log 2>&1) || tail -5 /tmp/mk.log; }
trap restore EXIT
echo "base"; run
for V in "$@"; do
  if (cd keds_amd/csrc; make -j8 EXTRA="$V" > /tmp/mk.log 2>&1); then echo "+ $V"; run; else echo "BUILD FAILED, skipped: $V"; tail -5 /tmp/mk.log; fi
done
restore
trap - EXIT
echo "base again"; run
