"""N > 1: a hung collective must cost a block of the line, not the line.  Every measurement behind the main one runs as a
PHASE with its own deadline; a daemon thread on every rank watches it.  When a phase overruns, rank 0 prints the line as far
as it got (the overrun phase and everything not reached recorded as {"error": ...}) and every rank leaves with os._exit(0):
a process group whose RCCL kernels hang cannot be torn down politely."""
import json
import os
import sys
import threading
import time

WATCH = {"line": None, "phase": None, "deadline": None, "rank": 0, "fired": False, "pending": []}


def _watchdog_loop():
    while True:
        time.sleep(0.5)
        dl = WATCH["deadline"]
        if dl is None or time.perf_counter() < dl:
            continue
        WATCH["fired"] = True
        if WATCH["rank"] == 0 and WATCH["line"] is not None:
            line = dict(WATCH["line"])
            msg = "watchdog: phase '%s' exceeded its %.0f s" % (WATCH["phase"], WATCH.get("budget", 0.0))
            line["watchdog"] = {"error": msg, "not_reached": list(WATCH["pending"])}
            sys.stdout.write(json.dumps(line) + "\n")
            sys.stdout.flush()
        elif WATCH["rank"] == 0 and not WATCH.get("printed"):
            sys.stderr.write("bench.py watchdog: phase '%s' overran before the main measurement was complete\n" % WATCH["phase"])
        os._exit(0 if (WATCH["line"] is not None or WATCH.get("printed")) else 3)


def watch_start(rank):
    WATCH["rank"] = rank
    threading.Thread(target=_watchdog_loop, daemon=True).start()


def arm(name, seconds):
    """a deadline without a `with` block (the main measurement spans several sections of bench.py)"""
    WATCH["phase"], WATCH["budget"] = name, seconds or 0.0
    WATCH["deadline"] = None if not seconds else time.perf_counter() + seconds


def disarm():
    WATCH["deadline"] = None


def done(name):
    WATCH["pending"] = [x for x in WATCH["pending"] if x != name]


class phase:
    """with phase("name", seconds): ... -- the block's deadline for the watchdog (None: no deadline)"""

    def __init__(self, name, seconds):
        self.name, self.seconds = name, seconds

    def __enter__(self):
        arm(self.name, self.seconds)
        self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        WATCH["deadline"] = None
        WATCH.setdefault("phase_seconds", {})[self.name] = round(time.perf_counter() - self.t0, 2)
        return False
