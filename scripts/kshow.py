#!/usr/bin/env python3
"""print name / calls / average us of a *_kernel_stats.csv of scripts/kstats.py"""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print('%-60s n=%3s avg %9s us' % (r['Name'][:60], r['FullSizeCalls'], r['AverageUs']))
