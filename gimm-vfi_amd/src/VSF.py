"""Vimeo-Septuplet-Flow benchmark of the motion-only model GIMM -- drop-in for reference src/VSF.py (same flags):
flows im1<->im7 through GIMM at the five interior frames, scored like src/VTF.py (reference VSF.py:61-170).
The protocol table and the evaluation loop live in VTF.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import VTF  # noqa: E402


def main(argv=None):
    return VTF.main(argv, protocol=VTF.SEPTUPLET, default_root="data/vimeo90k/vimeo_septuplet")


if __name__ == "__main__":
    main()
