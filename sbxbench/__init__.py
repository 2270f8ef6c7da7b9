"""The legs of bench.py (the CLI at the repo root): n1, dist, emulate, pmc, cpu, common."""
