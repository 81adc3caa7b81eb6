"""Host-side helpers of the reference-shaped package (setup / padding / flow visualisation)."""
