#!/bin/bash
# kernel sequence of one PlayLMP.training_step at B=32 (graph replay, serialised by the trace)
export TMPDIR=/tmp
O=gpurun_out/r4plseq; mkdir -p $O
cat > /tmp/run_pl.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
from tacorl_amd import _lib
dev = torch.device("cuda:0"); _lib.call("tacorl_hip_init", 0)
from tacorl_amd.modules.play_lmp.play_lmp_for_rl import PlayLMP
cams = ["rgb_static"]; T = 16
actor = {"policy": {"num_layers": 3, "hidden_dim": 256}}
pr = dict(num_heads=8, num_layers=2, encoder_hidden_size=2048, fc_hidden_size=4096, latent_plan_dim=16, min_std=1e-4, dropout_p=0.0, max_position_embeddings=T)
ad = dict(n_mixtures=10, num_layers=2, hidden_size=2048, out_features=7, num_classes=10, latent_plan_dim=16, rnn_model="rnn_decoder", include_goal=False)
torch.manual_seed(0)
p = PlayLMP(plan_proposal=actor, plan_recognition=pr, action_decoder=ad, plan_proposal_obs_modalities=cams, plan_proposal_goal_modalities=cams,
            plan_recognition_modalities=cams, action_decoder_modalities=cams, real_world=True, device=dev, compute_dtype="bf16", image_dtype="bf16")
batch = bench.synth_batch(int(os.environ.get("B", 32)), T, 84, 84, dev, 1)
p.enable_graph(); p.log_every_n_steps = 50
for _ in range(170): p.training_step(batch, 0)
torch.cuda.synchronize()
PY
B=${B:-32} timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 /tmp/run_pl.py > $O/run.log 2> $O/trace.err
python scratch/step_sequence.py $O/trace > $O/seq.txt
rm -rf $O/trace
tail -3 $O/seq.txt
