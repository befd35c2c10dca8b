"""TEST INFRASTRUCTURE ONLY: CPU oracle of the rain-streak hot path (see oracle/render.py).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the
product package (rain-rendering_amd/) never imports it and has no CPU fallback."""
