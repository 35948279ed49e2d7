// IO.cpp -- tool::ReadImageSequence / ReadImageSequenceWithPose: the reference's sequence directory format.
#include "Tool/IO.h"

#include <fstream>
#include <iostream>
#include <sstream>

namespace one_piece {
namespace tool {

void ReadImageSequence(const std::string& path, std::vector<std::string>& rgb_files, std::vector<std::string>& depth_files) {
    std::ifstream in((path + "/associate.txt").c_str());
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream fields(line);
        std::string t_rgb, rgb, t_depth, depth;
        fields >> t_rgb >> rgb >> t_depth >> depth; // a short line yields empty names, as the reference's parser does
        rgb_files.push_back(path + "/" + rgb);
        depth_files.push_back(path + "/" + depth);
    }
    std::cout << GREEN << "[ReadImageSequence]::[INFO]::Read " << rgb_files.size() << " images successfully." << RESET << std::endl;
}

void ReadImageSequenceWithPose(const std::string& path, std::vector<std::string>& rgb_files, std::vector<std::string>& depth_files,
                               std::vector<geometry::TransformationMatrix>& poses) {
    std::ifstream in((path + "/trajectory.txt").c_str());
    if (!in) {
        std::cout << RED << "[ReadImageSequenceWithPose]::[ERROR]::No file named trajectory.txt." << RESET << std::endl;
        return;
    }
    ReadImageSequence(path, rgb_files, depth_files);
    std::string line;
    geometry::TransformationMatrix pose; // a short line keeps the previous line's trailing entries, like the reference's reused matrix
    while (std::getline(in, line)) {
        std::istringstream fields(line);
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) fields >> pose(r, c);
        poses.push_back(pose);
    }
    if (poses.size() != rgb_files.size())
        std::cout << YELLOW << "[ReadImageSequenceWithPose]::[WARNING]:: The number of images and poses do not match." << RESET << std::endl;
}

} // namespace tool
} // namespace one_piece
