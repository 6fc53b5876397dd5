import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG = "3d_reconstruction_system_amd"


def r3d():
    return importlib.import_module(PKG)
