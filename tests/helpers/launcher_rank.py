"""Rank program for the launcher-supervision test: the rank named in argv[1] exits with code 3 at once, the
others wait (for a peer that will never come) until they are terminated."""
import os
import sys
import time

if os.environ["RANK"] == sys.argv[1]:
    sys.stderr.write("this rank fails on purpose\n")
    sys.exit(3)
if os.environ["RANK"] == "0":
    print("rank 0 started", flush=True)
time.sleep(float(sys.argv[2]) if len(sys.argv) > 2 else 120)
