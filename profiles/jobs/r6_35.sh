#!/bin/bash
timeout 300 python scripts/act_chain_probe.py 2>&1 | grep -v Warning | tail -10
