"""Maximising algebraic connectivity (MAC) candidate sparsifier -- restatement of the
algorithm vendored by the reference under cslam/mac/ (MarineRoboticsGroup/mac)."""
