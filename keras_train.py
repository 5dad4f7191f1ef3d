#!/usr/bin/env python3
"""Entry point kept under the reference's script name so `make train` is unchanged (reference Makefile:34-61)."""
from k210_yolo_framework_amd.training import cli

if __name__ == '__main__':
    cli()
