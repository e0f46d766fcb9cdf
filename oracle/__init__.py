"""Parity oracle (CPU restatement + optional compiled reference) -- TEST INFRASTRUCTURE ONLY."""
