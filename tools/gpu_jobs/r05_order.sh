#!/bin/bash
bash tools/gpu_jobs/ab_libs.sh 4 c2 main dwolast
