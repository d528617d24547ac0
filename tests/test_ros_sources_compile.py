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


def test_catkin_package_files_configure_and_build_in_the_ros_free_mode(tmp_path):
    """ros/CMakeLists.txt + ros/package.xml (the catkin package a ROLO maintainer builds, round 4's verdict item 6): configured with -DROLO_HIP_MOCK_ROS=ON — the
    branch that replaces find_package(catkin) by the stand-in headers — all four node targets build (-Wall -Wextra -Werror) and link against librolo_hip.so.
    The catkin branch itself (catkin_package, ${catkin_LIBRARIES}, the message-generation dependency) cannot run here: no ROS in the image."""
    import shutil
    import xml.etree.ElementTree as ET
    import rolo_amd.build as B
    B.build()
    cmake = shutil.which("cmake")
    if not cmake:
        pytest.skip("no cmake on PATH")
    r = subprocess.run([cmake, "-DROLO_HIP_MOCK_ROS=ON", os.path.join(ROOT, "ros")], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run([cmake, "--build", ".", "--parallel", "4"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    for node in NODES:
        assert os.path.exists(tmp_path / node)
    # without ROS and without the mock switch the configuration must stop with the instruction, not half-configure
    other = tmp_path / "nomock"; other.mkdir()
    r = subprocess.run([cmake, os.path.join(ROOT, "ros")], cwd=other, capture_output=True, text=True)
    assert r.returncode != 0 and "ROLO_HIP_MOCK_ROS" in (r.stdout + r.stderr)
    # the manifest names the package of the CMake project and every package the node sources include messages from
    pkg = ET.parse(os.path.join(ROOT, "ros", "package.xml")).getroot()
    deps = {d.text for d in pkg.findall("depend")}
    assert pkg.find("name").text == "rolo_hip_nodes" and {"roscpp", "tf", "rolo", "sensor_msgs", "nav_msgs", "autoware_rviz_msgs"} <= deps
    txt = open(os.path.join(ROOT, "ros", "CMakeLists.txt")).read()
    assert "project(rolo_hip_nodes" in txt and all(n in txt for n in NODES)
