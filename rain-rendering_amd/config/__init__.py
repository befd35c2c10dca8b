"""Dataset plug-ins with the reference's interface (reference README.md:176-178):
each module exposes resolve_paths(params) and settings()."""
