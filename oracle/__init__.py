"""CPU oracle for the YACHT hot path — TEST INFRASTRUCTURE ONLY (see oracle/README.md)."""
