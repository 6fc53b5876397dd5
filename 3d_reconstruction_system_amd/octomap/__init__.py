"""Drop-in counterparts of the reference's octomap/ scripts (txt / PLY cloud -> OctoMap .bt)."""
