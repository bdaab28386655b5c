"""CPU oracle: TEST INFRASTRUCTURE ONLY (see oracle/oracle.h).  Nothing in the product imports it."""
