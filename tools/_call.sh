M="python tools/model_bench.py --bf16 --act-bf16 --no-events --steps 20 --warmup 5"
ms() { python -c "import sys,json; print('$1', json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])['ms_per_step_bare'])"; }
echo "# config 4 (ResidualUNet3D f_maps=64, 1x80x160x160, bf16 + bf16 storage), ms per fwd+bwd step, bare timing, ONE box, interleaved"
echo "# old = U3D_WGRAD_JOB=0 U3D_PACK_BOTH=0 U3D_CKPT_RERUN_LAST=1 (the three host-selectable changes of the last session off; the kernel-level ones stay)"
for i in 1 2 3; do
U3D_WGRAD_JOB=0 U3D_PACK_BOTH=0 $M 2>/dev/null | ms "bare old"
$M 2>/dev/null | ms "bare new"
U3D_WGRAD_JOB=0 U3D_PACK_BOTH=0 U3D_CKPT_RERUN_LAST=1 $M --checkpoint 2>/dev/null | ms "ckpt old"
$M --checkpoint 2>/dev/null | ms "ckpt new"
U3D_WGRAD_JOB=0 U3D_PACK_BOTH=0 U3D_CKPT_RERUN_LAST=1 $M --checkpoint --checkpoint-levels 2 2>/dev/null | ms "ckpt2 old"
$M --checkpoint --checkpoint-levels 2 2>/dev/null | ms "ckpt2 new"
done
