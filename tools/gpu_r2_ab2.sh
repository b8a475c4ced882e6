#!/bin/bash
AB_CONFIG=4 AB_LANES=32768 bash tools/gpu_ab.sh base sleep8
AB_CONFIG=5 AB_LANES=65536 bash tools/gpu_ab.sh base sleep8
AB_CONFIG=2 AB_LANES=1024 bash tools/gpu_ab.sh base sleep8
