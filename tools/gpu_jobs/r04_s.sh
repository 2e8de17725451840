#!/bin/bash
tools/gpu_jobs/ab_libs.sh 1 c2 main pl_NODMA pl_NOSTORE pl_NOMMA pl_ALL main
