#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4pr; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "plan_recognition" > $O/k.txt 2>&1; echo "rc=$?" >> $O/k.txt
timeout 1500 python -m pytest tests/test_step_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "playlmp or attention or layernorm or plan_rec or transformer" > $O/t.txt 2>&1; echo "rc=$?" >> $O/t.txt
python - > $O/time.txt 2>&1 <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
import bench
from tacorl_amd import _lib
from tacorl_amd.modules.play_lmp.play_lmp_for_rl import PlayLMP
dev = torch.device("cuda:0"); _lib.call("tacorl_hip_init", 0)
actor = {"policy": {"num_layers": 3, "hidden_dim": 256}}
pr = dict(num_heads=8, num_layers=2, encoder_hidden_size=2048, fc_hidden_size=4096, latent_plan_dim=16, min_std=1e-4, dropout_p=0.0, max_position_embeddings=16)
ad = dict(n_mixtures=10, num_layers=2, hidden_size=2048, out_features=7, num_classes=10, latent_plan_dim=16, rnn_model="rnn_decoder", include_goal=False)
cams = ["rgb_static"]
for B in (32, 256):
    torch.manual_seed(0)
    m = PlayLMP(plan_proposal=actor, plan_recognition=pr, action_decoder=ad, plan_proposal_obs_modalities=cams, plan_proposal_goal_modalities=cams,
                plan_recognition_modalities=cams, action_decoder_modalities=cams, real_world=True, device=dev, compute_dtype="bf16", image_dtype="bf16")
    batch = bench.synth_batch(B, 16, 84, 84, dev, 1)
    m.enable_graph(); m.log_every_n_steps = 50
    out = {}
    for rep in range(2):
        for fused in (False, True):
            m.pr.fused_train = fused; m._graphs = {}
            for _ in range(8): m.training_step(batch, 0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(200): m.training_step(batch, 0)
            torch.cuda.synchronize(); out.setdefault(fused, []).append((time.perf_counter() - t0) / 200 * 1e3)
    print(f"PlayLMP B={B}: per-op forward {min(out[False]):.4f} ms/step, fused train forward {min(out[True]):.4f} ms/step", flush=True)
    m._graphs = {}; del m; torch.cuda.synchronize(); torch.cuda.empty_cache()
PY
tail -n 5 $O/k.txt $O/t.txt; cat $O/time.txt
