"""Input synthesis for benchmarks and tests (not part of the query path)."""
