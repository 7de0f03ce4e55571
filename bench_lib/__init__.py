"""Parts of bench.py (the one file the driver runs): workloads and byte accounting, the oracle legs, the watchdog, the N > 1
gather measurements, the BASELINE C5 block and the secondary workloads of the default line.  bench.py keeps argument parsing,
the main timed regions and the JSON line."""
