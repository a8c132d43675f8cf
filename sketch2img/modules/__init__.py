"""Alias package: `modules.X` resolves to sketch2img_amd.modules.X (drop-in for the reference layout)."""
