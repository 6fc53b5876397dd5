#!/usr/bin/env python3
"""The reference ships other_tools/ply_transfer_octomap.py as a byte-identical copy of
octomap/ply_transfer_octomap.py; this one simply forwards to the drop-in."""
import os
import sys

if __package__ in (None, ""):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "octomap"))
    from ply_transfer_octomap import *  # type: ignore  # noqa: F401,F403
    from ply_transfer_octomap import main  # type: ignore
else:
    from ..octomap.ply_transfer_octomap import *  # noqa: F401,F403
    from ..octomap.ply_transfer_octomap import main

if __name__ == '__main__':
    main()
