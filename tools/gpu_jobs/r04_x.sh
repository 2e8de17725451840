#!/bin/bash
PYTHONPATH=$PWD python tools/debug/adamw_time.py 2>&1 | grep -v amdgpu.ids
