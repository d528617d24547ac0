"""CPU: the catkin node sources under ros/ compile (-Wall -Wextra -Werror) and link against librolo_hip.so when the roscpp API they
touch is declared by the stand-in headers of tests/cpp/mock_ros (this image has no ROS). A type check of ros/*.cpp and
ros/rolo_ros_convert.hpp — names, message fields, the node-core interfaces of include/rolo_ros_nodes.hpp — not a test of ROS transport."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODES = ["rolo_imageProjection_hip", "rolo_featureExtraction_hip", "rolo_lidarOdometry_hip", "rolo_front_fused_hip"]


@pytest.mark.parametrize("node", NODES)
def test_node_source_compiles_and_links(node, tmp_path):
    import rolo_amd.build as B
    B.build()   # librolo_hip.so (cross-compiled for gfx950; no GPU needed to link against it)
    exe = str(tmp_path / node)
    cmd = ["g++", "-std=c++17", "-O0", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "tests", "cpp", "mock_ros"), "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "ros"), os.path.join(ROOT, "ros", node + ".cpp"), "-o", exe,
           "-L", os.path.join(ROOT, "rolo_amd"), "-lrolo_hip", "-Wl,-rpath," + os.path.join(ROOT, "rolo_amd"), "-Wl,-rpath,/opt/rocm/lib", "-pthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    assert os.path.exists(exe)


def test_pcl_branch_of_the_drop_in_class_type_checks(tmp_path):
    """include/rot_vgicp_hip.hpp with ROLO_HIP_WITH_PCL against tests/cpp/mock_pcl: every member function instantiated"""
    import rolo_amd.build as B
    B.build()
    exe = str(tmp_path / "shim_pcl_check")
    cmd = ["g++", "-std=c++17", "-O0", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "tests", "cpp", "mock_pcl"), "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "shim_pcl_check.cpp"), "-o", exe,
           "-L", os.path.join(ROOT, "rolo_amd"), "-lrolo_hip", "-Wl,-rpath," + os.path.join(ROOT, "rolo_amd"), "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
