#!/bin/bash
timeout -k 10 200 python tools/plan_marks.py 4 vit-b16 2>/dev/null | grep -v "^ *f[0-9]" 
timeout -k 10 200 python tools/plan_marks.py 2 vit-b16 2>/dev/null | grep "plan\|tail\|begin\|head"
