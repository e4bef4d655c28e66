"""Offline helpers of the reference: the two BAM down-samplers used for the
titration experiments and the theoretical limit-of-detection script.  Host-only Python; nothing here is on
the device path."""
