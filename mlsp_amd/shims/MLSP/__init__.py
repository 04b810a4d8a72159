"""Import shim: `from MLSP import PCM, mlsp` (PointDA/trainer.py:15): the HIP loss module, the on-device input corruption
(mlsp.deform_input) and the mixup with on-device farthest point sampling (PCM.mix_shapes)."""
