#!/bin/bash
for v in 400 402 404 800 802 804 1600 1602 1604 1802 1804 2602 1402; do LRAM_READ_VARIANT=$v python scripts/read_ceiling.py; done
WITH_COPY=1 python scripts/read_ceiling.py
