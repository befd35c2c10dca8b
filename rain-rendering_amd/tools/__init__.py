"""Counterparts of the reference's tools/ (particle simulation)."""
