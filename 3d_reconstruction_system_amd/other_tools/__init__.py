"""Drop-in counterparts of the reference's other_tools/transfer_T_icp.py (same function names, argument
meaning, default paths and file formats); the per-point work runs on the MI355X."""
