// Tool/IO.h -- the sequence-directory readers the fusion drivers call (reference: src/Tool/IO.cpp:59-108):
// associate.txt ("t_rgb rgb_path t_depth depth_path" per line) and trajectory.txt (16 floats per line = row-major
// camera-to-world pose).
#pragma once
#include <string>
#include <vector>

#include "Geometry/Geometry.h"

namespace one_piece {
namespace tool {

void ReadImageSequence(const std::string& path, std::vector<std::string>& rgb_files, std::vector<std::string>& depth_files);
void ReadImageSequenceWithPose(const std::string& path, std::vector<std::string>& rgb_files, std::vector<std::string>& depth_files,
                               std::vector<geometry::TransformationMatrix>& poses);

} // namespace tool
} // namespace one_piece
