"""Synthetic workload generators shared by the benches and the checker (no arithmetic of the path lives here)."""
