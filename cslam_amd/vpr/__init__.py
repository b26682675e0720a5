"""Visual place recognition extractors (reference: cslam/vpr/)."""
