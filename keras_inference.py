#!/usr/bin/env python3
"""Entry point kept under the reference's script name so `make inference` is unchanged (Makefile:65-76)."""
from k210_yolo_framework_amd.inference import cli

if __name__ == '__main__':
    cli()
