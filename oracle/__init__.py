"""CPU oracle of the hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product path (glorie_slam_amd/) never does."""
