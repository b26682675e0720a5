"""Lidar place recognition on the MI355X (drop-in for cslam/lidar_pr/scancontext_matching.py)."""
