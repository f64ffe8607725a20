#!/usr/bin/env python
"""Wrapper: the compress harness lives in efficient-speech-codec_amd/scripts/compress.py (python -m scripts.compress)."""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "efficient-speech-codec_amd"))
runpy.run_module("scripts.compress", run_name="__main__")
