#!/bin/bash
AECF_LIB_PATH=$PWD/build/var/wtl/libaecf_hip.so PYTHONPATH=$PWD python tools/debug/ws_timeline.py 2>&1 | grep -v amdgpu.ids | cut -c1-330
