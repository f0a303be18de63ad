"""`loco_rl` import surface for unmodified reference scripts (`from loco_rl.runners import OnPolicyRunner`,
locotouch/scripts/train.py:128, locotouch/distill/distillation.py:17).  Registered in `sys.modules` under the name
`loco_rl` by locotouch_amd.compat.runtime.install(); every name resolves to the PyTorch-ROCm trainer in locotouch_amd.rl."""
