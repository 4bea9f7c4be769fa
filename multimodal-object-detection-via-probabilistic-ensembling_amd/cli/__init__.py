"""Command-line counterparts of the reference's demo scripts (demo/FLIR/*.py, demo/KAIST/demo_LAMR_KAIST.py)."""
