#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for cfg in "64 64" "64 4"; do
set -- $cfg
for act in 0 1; do
DBG_ACT=$act TACORL_SCRATCH_LIB=scratch/libs/ef_nospread.so python scratch/r6_dbg_spread.py /tmp/b.pt $1 $2
DBG_ACT=$act python scratch/r6_dbg_spread.py /tmp/a.pt $1 $2
python - <<PY
import torch
(oa, aa, lay), (ob, ab, _) = torch.load("/tmp/a.pt"), torch.load("/tmp/b.pt")
d = (oa[0] - ob[0]).abs().amax(dim=1)
print("cfg $1 wg $2 act $act: out images differing", int((d > 0).sum()), "of", oa[0].shape[0], "max", float(d.max()), "ref max", float(ob[0].abs().max()))
if aa[0] is not None:
    offs, tot = lay[0]
    names = ["y1", "y2", "y3", "softargmax", "fc1"]
    for j, nm in enumerate(names):
        end = offs[j + 1] if j + 1 < 5 else tot
        x, y = aa[0][offs[j]:end], ab[0][offs[j]:end]
        nz = (x != y).nonzero().flatten()
        print("   ", nm, "elements differing", nz.numel(), "of", x.numel(), "first", nz[:6].tolist(), "max", float((x - y).abs().max()) if nz.numel() else 0.0)
PY
done
done
