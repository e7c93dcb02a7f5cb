// footage.hpp — the capture's .bin container (BinaryFootageFile.cpp / .h): a 4096-byte metadata page {magic 0xfaceb00c,
// timestamp, fileIndex, fileCount, width, height, bitsPerPixel, numberOfCameras}, then frames interleaved by camera, 8 or 12
// bits per pixel packed; the camera's serial number is the second 32-bit word of every frame. Read-only mmap. Shared by
// host/Unpacker and host/TestRenderStereoPanorama --bin_list.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>

namespace footage {
struct Header { uint32_t magic, timestamp, fileIndex, fileCount, width, height, bitsPerPixel, numberOfCameras; };
struct Footage {
  std::string path;
  int fd = -1;
  const uint8_t* base = nullptr;
  size_t size = 0;
  Header md{};
  size_t frame_size() const { return (size_t)md.width * md.height * md.bitsPerPixel / 8; }
  size_t frames() const { return (md.numberOfCameras && frame_size() && size >= 4096) ? (size - 4096) / frame_size() / md.numberOfCameras : 0; }
  const uint8_t* frame(size_t f, size_t cam) const {  // (offsets, not pointers: nothing here may wrap around)
    const size_t fs = frame_size();
    if (fs < 8) throw std::runtime_error("no frames (bits per pixel / sizes of the header) in " + path);  // (the serial number is bytes 4..7)
    const size_t avail = (size - 4096) / fs;  // whole frames in the file
    if (cam >= md.numberOfCameras || f >= avail / md.numberOfCameras) throw std::runtime_error("frame out of range for " + path);
    return base + 4096 + (md.numberOfCameras * f + cam) * fs;
  }
  void open(bool verbose = true) {
    fd = ::open(path.c_str(), O_RDONLY);
    if (fd == -1) throw std::runtime_error("Error opening file " + path + ": " + std::strerror(errno));
    struct stat st;
    if (fstat(fd, &st) < 0) throw std::runtime_error("Error retrieving stat() information for file " + path);
    size = (size_t)st.st_size;
    if (size < 4096) throw std::runtime_error("not a footage file (shorter than its metadata page): " + path);
    void* a = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (a == MAP_FAILED) throw std::runtime_error("Error mmap'ing() file " + path);
    base = static_cast<const uint8_t*>(a);
    std::memcpy(&md, base, sizeof md);
    // an untrusted header: sizes that no sensor has would wrap frame_size() around (BinaryFootageFile.cpp trusts them)
    if (md.numberOfCameras != 0) {
      if (md.width == 0 || md.height == 0 || md.width > 65536u || md.height > 65536u || md.numberOfCameras > 4096u)
        throw std::runtime_error("implausible metadata (width / height / numberOfCameras) in " + path);
      if (md.bitsPerPixel != 8 && md.bitsPerPixel != 12) throw std::runtime_error("unsupported bits per pixel (8 and 12 exist) in " + path);
      if (md.bitsPerPixel == 12 && (md.width & 1u)) throw std::runtime_error("12-bit frames need an even width: " + path);
      if (frame_size() < 8) throw std::runtime_error("implausible metadata (a frame of fewer than 8 bytes) in " + path);
    }
    if (verbose) std::printf("Metadata:\nmagic = %x\ntimestamp = %u\nfileIndex = %u\nfileCount = %u\nwidth = %u\nheight = %u\nbpp = %u\nnumberOfCameras = %u\n",
                md.magic, md.timestamp, md.fileIndex, md.fileCount, md.width, md.height, md.bitsPerPixel, md.numberOfCameras);
  }
  ~Footage() {
    if (base) munmap(const_cast<uint8_t*>(base), size);
    if (fd != -1) ::close(fd);
  }
};
}  // namespace footage
