// rosbag_reader.h -- reads sensor_msgs/PointCloud2 messages of one topic out of a rosbag
// (format "#ROSBAG V2.0") without ROS: what input_data_node does with rosbag::Bag / rosbag::View
// and pcl::fromROSMsg (input_data.cpp:160-190, 305-313; scanRegistration.cpp:262-264).
#ifndef GPSCAL_ROSBAG_READER_H
#define GPSCAL_ROSBAG_READER_H

#include <string>
#include <vector>

namespace gpscal_host {

struct CloudSeries {
    std::vector<float> xyz;       // packed x, y, z of every point of every message
    std::vector<int> sweep_off;   // messages + 1 point offsets (starts as {0})
    std::vector<double> stamps;   // header.stamp.toSec() per message
};

// Appends the topic's PointCloud2 messages of `path`, ordered by their record time, to `out`.
// The topic matches with or without a leading slash.  Chunks may be uncompressed, bz2 or lz4 (libbz2 /
// liblz4 are loaded at run time when present).  Returns false and
// fills `err` on any malformed record.
bool read_bag_clouds(const std::string &path, const std::string &topic, CloudSeries &out, std::string &err);

}  // namespace gpscal_host
#endif
