"""Drop-in counterparts of the reference's transfer/ scripts (same function names, argument
meaning, default paths and file formats); the per-point work runs on the MI355X."""
