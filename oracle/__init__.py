"""ORACLE -- test infrastructure only (see crfp_oracle.py header). Never imported by crfp_amd."""
