#!/bin/bash
timeout 300 python scripts/act_host_profile.py 2>&1 | grep -v Warning | cut -c1-170 | head -90
