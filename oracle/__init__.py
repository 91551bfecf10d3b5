"""CPU oracle for the MSM/NTT hot path -- TEST INFRASTRUCTURE ONLY (see pyref.py / oracle.c)."""
