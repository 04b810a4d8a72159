"""Import shim: `from MLSP import PCM, mlsp` (PointDA/trainer.py:15).  `mlsp` is the HIP loss module with the
host-side helpers of the reference attached when the reference package is importable; PCM is the reference's own."""
