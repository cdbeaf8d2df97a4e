#pragma once
// radx_device.hpp — radx::PhysicalDeviceHelper and radx::Device over a HIP device ordinal
// (reference: include/radx/radx_device.hpp:12-138, radx_implement.inl:197-311).
// The reference's Device owns a VMA allocator, a descriptor pool, a pipeline cache and two
// descriptor-set layouts; none of those exist in HIP, so Device is the ordinal plus its
// properties, and "initialize" makes sure the ordinal is usable.
#include "radx_core.hpp"

namespace radx {

    class PhysicalDeviceHelper : public std::enable_shared_from_this<PhysicalDeviceHelper> {
    protected:
        PhysicalDevice physicalDevice{};
        hipDeviceProp_t properties{};
        bool haveProperties = false;
        friend class Device;

    public:
        // reference: radx_device.hpp:38-40 (query features/properties, pick the shader directory)
        PhysicalDeviceHelper(const PhysicalDevice& physicalDevice) : physicalDevice(physicalDevice) {
            haveProperties = hipGetDeviceProperties(&properties, physicalDevice.ordinal) == hipSuccess;
        }

        // reference: radx_device.hpp:51 returns the SPIR-V directory name ("amdgpu", "turing", ...);
        // there is one code object here, named after the ISA it was compiled for.
        std::string getDriverName() const { return haveProperties ? std::string(properties.gcnArchName) : std::string("gfx950"); }
        // reference: radx_device.hpp:53-60; a CDNA wavefront is 64 lanes.
        uint32_t getRecommendedSubgroupSize() const { return 64u; }
        bool available() const { return haveProperties; }
        const hipDeviceProp_t& getProperties() const { return properties; }

        operator PhysicalDevice&() { return physicalDevice; }
        operator const PhysicalDevice&() const { return physicalDevice; }
    };

    class Device : public std::enable_shared_from_this<Device> {
    protected:
        LogicalDevice device{};
        std::shared_ptr<PhysicalDeviceHelper> physicalHelper{};

    public:
        ~Device() {
            // reference: radx_device.hpp:86-89 waits for the device to go idle before teardown
            if (physicalHelper && physicalHelper->available()) {
                int prev = -1;
                if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(device.ordinal) == hipSuccess) {
                    (void)hipDeviceSynchronize();
                    if (prev >= 0) (void)hipSetDevice(prev);
                }
            }
        }

        // reference: radx_device.hpp:95, radx_implement.inl:197-311
        std::shared_ptr<Device> initialize(const LogicalDevice& device, std::shared_ptr<PhysicalDeviceHelper> physicalHelper) {
            this->device = device;
            this->physicalHelper = physicalHelper;
            check_hip(hipSetDevice(device.ordinal), "hipSetDevice");
            return shared_from_this();
        }

        int ordinal() const { return device.ordinal; }
        std::shared_ptr<PhysicalDeviceHelper>& getPhysicalHelper() { return physicalHelper; }
        const std::shared_ptr<PhysicalDeviceHelper>& getPhysicalHelper() const { return physicalHelper; }
        std::string getDriverName() const { return physicalHelper->getDriverName(); }
        uint32_t getRecommendedSubgroupSize() const { return physicalHelper->getRecommendedSubgroupSize(); }

        operator LogicalDevice&() { return device; }
        operator const LogicalDevice&() const { return device; }
    };
}
