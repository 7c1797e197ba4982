"""Measurement helpers (none of them on the product path).  A regular package, so that `import tools.bench_roofline` from bench.py resolves HERE even if some
site-packages distribution ships a top-level `tools` package (a namespace package would lose to it)."""
