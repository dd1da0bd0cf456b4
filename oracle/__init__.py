"""Test infrastructure only: CPU oracle (restatement) and partial reference build. Never imported by classpro_amd/."""
