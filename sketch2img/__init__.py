"""Alias package: the reference also imports `sketch2img.modules.X` (modules/clip_guided_inf.py:7)."""
