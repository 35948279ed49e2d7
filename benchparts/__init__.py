"""The supplementary sections of bench.py's JSON line, one module each: run(c, out) fills its keys of `out`."""
