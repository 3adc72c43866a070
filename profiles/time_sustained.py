#!/usr/bin/env python3
"""SUSTAINED tick time of the streamed per-tick path: 0.6 s of back-to-back ticks over a long generated input
sequence (about 6 GB resident, like bench.py), reported chunk by chunk so transients show.

    [QLE_NT=0|1|2] [QLE_REFRESH=R] python profiles/time_sustained.py <batch> predict|mixed

`mixed` = the cfg3 schedule (every 14th tick fused).  Found with this script (profiles/r01_tuning.md section 5): with
non-temporal loads AND stores on every tick a 36 MiB state starts at 9.1 us per predict and decays to 10.9 us within
~3 000 ticks, so numbers taken in the first 40 ms of a process were optimistic."""
