#!/usr/bin/env python3
"""the cfg3 losses leg for a kernel trace: python3 tools/prof_cfg3.py"""
import os
import sys
sys.argv = [sys.argv[0], 'cfg3_losses']
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profile_leg.py')).read())
