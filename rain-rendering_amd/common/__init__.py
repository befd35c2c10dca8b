"""Host-side mirror of the reference's ``common`` package for the hot path."""
