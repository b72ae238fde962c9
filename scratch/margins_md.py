"""gpurun_out/parity_margins.jsonl (written by the GPU tests, tests/golden_util.record_margin) -> a markdown report:
per test the base tolerance, how many tensors were compared, how many were held to a WIDENED tolerance (3 x the tensor's
own reproducibility floor) and every widened or tight (> 50 % of its tolerance) comparison in full."""
import collections
import json
import sys

rows = [json.loads(l) for l in open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/parity_margins.jsonl")]
by = collections.OrderedDict()
for r in rows:  # the worst occurrence of a (test, kind, tensor) over steps / re-runs
    k = (r["test"], r["kind"], r["tensor"])
    if k not in by or r["err"] / max(r["tol"], 1e-30) > by[k]["err"] / max(by[k]["tol"], 1e-30):
        by[k] = r
tests = collections.OrderedDict()
for (t, kind, tensor), r in by.items():
    tests.setdefault(t, []).append((kind, tensor, r))
print("# Parity margins of the whole-step GPU tests (generated: scratch/margins_md.py from the tests' own comparisons)\n")
print("Every comparison a step test makes against the oracle is logged (tests/golden_util.record_margin): `err` measured, `tol`\n"
      "the tolerance it was held to (base; max(base, 3 x floor) only for a tensor that missed base), `floor` = the tensor's own reproducibility (its change under a 1-ulp\n"
      "perturbation of the parameters, for bf16 also half of what rounding operands to bf16 changes at all -\n"
      "golden_util.gradient_floor, test_step_gpu._with_bf16_sensitivity).  f32 tests: gradients base 3e-4 against the fp32\n"
      "oracle; bf16 tests: logged scalars / plans 2e-3, gradients 1e-2 against the oracle with the MFMA's operand rounding.\n"
      "Listed in full: every comparison with a widened tolerance or one that used more than half of its tolerance.\n")
print("| test | comparisons | base tol (grad) | widened | largest tol | worst err / tol |\n|---|---|---|---|---|---|")
detail = []
for t, lst in tests.items():
    grads = [r for k, _, r in lst if k == "grad"]
    base = min((r["tol"] for r in grads), default=float("nan"))
    wid = [(k, n, r) for k, n, r in lst if k == "grad" and r["tol"] > base * 1.0001]
    worst = max(lst, key=lambda x: x[2]["err"] / max(x[2]["tol"], 1e-30))
    print(f"| {t.split('::')[-1]} | {len(lst)} | {base:.0e} | {len(wid)} | {max((r['tol'] for r in grads), default=float('nan')):.2e} | "
          f"{100 * worst[2]['err'] / max(worst[2]['tol'], 1e-30):.0f} % ({worst[1]}) |")
    tight = [(k, n, r) for k, n, r in lst if r["err"] > 0.5 * r["tol"] and (k, n, r) not in wid]
    if wid or tight:
        detail.append((t, base, wid, tight))
for t, base, wid, tight in detail:
    print(f"\n## {t.split('::')[-1]}\n\n| kind | tensor | err | tol | floor | err / tol |\n|---|---|---|---|---|---|")
    for kind, tensor, r in sorted(wid + tight, key=lambda x: -x[2]["tol"]):
        fl = "" if r["floor"] is None else f"{r['floor']:.2e}"
        print(f"| {kind} | {tensor} | {r['err']:.2e} | {r['tol']:.2e}{' (widened)' if (kind, tensor, r) in wid else ''} | {fl} | "
              f"{100 * r['err'] / max(r['tol'], 1e-30):.0f} % |")

insitu = [(t, k, n, r) for t, lst in tests.items() for k, n, r in lst if k.startswith("encoder backward vs rounded oracle") or k.startswith("upstream")]
if insitu:
    print("\n## PlayLMP (C1, bf16): the encoder backward held to the rounded oracle on the step's OWN d_emb (nothing widened)\n\n"
          "| tensor | err | tol | note |\n|---|---|---|---|")
    for t, k, n, r in insitu:
        tol = "" if r["tol"] != r["tol"] else f"{r['tol']:.1e}"
        print(f"| {n} | {r['err']:.2e} | {tol} | {k} |")
