#!/bin/bash
# what is in the 67 us that remain with no MFMA, no fragment reads and no DMA?  (timing-only scratch builds)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
SPEC="4096 512a 256 512a 512a 512 512"
echo "product:                       $(python scratch/run_fused.py $SPEC | head -1)"
echo "p8  (no per-image barriers):   $(TACORL_SCRATCH_LIB=scratch/libs/ef_p8.so python scratch/run_fused.py $SPEC | head -1)"
echo "p16 (no soft-argmax math):     $(TACORL_SCRATCH_LIB=scratch/libs/ef_p16.so python scratch/run_fused.py $SPEC | head -1)"
echo "f0  (no mfma/reads/dma):       $(TACORL_SCRATCH_LIB=scratch/libs/ef_f0.so python scratch/run_fused.py $SPEC | head -1)"
echo "f8  (+ no barriers):           $(TACORL_SCRATCH_LIB=scratch/libs/ef_f8.so python scratch/run_fused.py $SPEC | head -1)"
echo "f16 (+ no soft-argmax):        $(TACORL_SCRATCH_LIB=scratch/libs/ef_f16.so python scratch/run_fused.py $SPEC | head -1)"
echo "f24 (+ neither):               $(TACORL_SCRATCH_LIB=scratch/libs/ef_f24.so python scratch/run_fused.py $SPEC | head -1)"
echo "f26 (+ no conv1 act1 stores):  $(TACORL_SCRATCH_LIB=scratch/libs/ef_f26.so python scratch/run_fused.py $SPEC | head -1)"
echo "plain problems only, product:  $(python scratch/run_fused.py 4096 1536 1280 | head -1)"
echo "plain problems only, f0:       $(TACORL_SCRATCH_LIB=scratch/libs/ef_f0.so python scratch/run_fused.py 4096 1536 1280 | head -1)"
