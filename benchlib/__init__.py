"""Parts of bench.py that are not the contract line: the reference-order secondaries (orders.py)."""
