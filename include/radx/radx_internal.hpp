#pragma once
// radx_internal.hpp — InternalInterface, InputInterface, Algorithm, Sort<T>, Radix with the
// reference's method names and call order (reference: include/radx/radx_internal.hpp:8-148,
// include/radx/radx_implement.inl:315-490), bound to the gfx950 engine through include/radx_hip.h.
//
//   auto helper  = std::make_shared<radx::PhysicalDeviceHelper>(radx::PhysicalDevice(0));
//   auto device  = std::make_shared<radx::Device>()->initialize(radx::LogicalDevice(0), helper);
//   auto program = std::make_shared<radx::Radix>(); program->initialize(device);
//   auto sorter  = std::make_shared<radx::Sort<radx::Radix>>(); sorter->initialize(device, program, n);
//   auto input   = std::make_shared<radx::InputInterface>(device);
//   input->setElementCount(n).setKeysBufferInfo(keys).setSwapBufferInfo(swap).buildDescriptorSet();
//   sorter->command(radx::CommandBuffer(stream), input);   // enqueues; result lands in `keys`
#include "radx_buffer.hpp"
#include "radx_device.hpp"

namespace radx {

    // Scratch owned by one Sort object: the engine handle with its histogram / prefix-scan tables
    // (reference: InternalInterface::bufferMemory, radx_internal.hpp:11, sized by
    // Radix::createInternalMemory, radx_implement.inl:450-490).
    class InternalInterface {
    protected:
        std::shared_ptr<Device> device{};
        radx_hip_sorter* sorter = nullptr;
        std::size_t maxElementCount = 1024 * 1024;
        int keyBytes = 4, valueBytes = 0;

    public:
        friend class Algorithm;
        friend class Radix;
        InternalInterface() {}
        InternalInterface(const std::shared_ptr<Device>& device) : device(device) {}
        InternalInterface(const InternalInterface&) = delete;
        InternalInterface& operator=(const InternalInterface&) = delete;
        virtual ~InternalInterface() { if (sorter) (void)radx_hip_destroy(sorter); }

        // Views of the scratch tables (reference: radx_internal.hpp:25-28).  histogramBufferInfo and
        // prefixScansBufferInfo are bindings 3 and 4 of the reference's shaders (counting.comp:45,
        // partition.comp:26): device memory holding counts[tile][16] and partitions[tile][16] of
        // the LAST pass of the last command(), refreshed by Radix::command (the engine rotates
        // its tables, so the addresses change from sort to sort).  prefixScansBufferInfo is the
        // complete table when prefixGroupBasesBufferInfo.buffer is null (inputs of up to 16384
        // tiles); for larger inputs the scan kernels leave it in two parts, see
        // radx_hip_device_tables.  The key staging views are the reference's 256-byte stubs
        // (radx_implement.inl:463-471): nothing reads them, here as there.
        DescriptorBufferInfo keysStoreBufferInfo{}, keysBackupBufferInfo{}, keysCacheBufferInfo{}, countMaximBufferInfo{},
            histogramBufferInfo{}, prefixScansBufferInfo{}, prefixGroupBasesBufferInfo{};
        std::size_t prefixTilesPerGroup = 0;  // tiles that share one row of prefixGroupBasesBufferInfo
        virtual InternalInterface& setKeysStoreBufferInfo(const DescriptorBufferInfo& keysStore = {}) { keysStoreBufferInfo = keysStore; return *this; }
        virtual InternalInterface& setKeysBackupBufferInfo(const DescriptorBufferInfo& keysBackup = {}) { keysBackupBufferInfo = keysBackup; return *this; }
        virtual InternalInterface& setKeysCacheBufferInfo(const DescriptorBufferInfo& keysCache = {}) { keysCacheBufferInfo = keysCache; return *this; }
        virtual InternalInterface& setCountMaximBufferInfo(const DescriptorBufferInfo& references = {}) { countMaximBufferInfo = references; return *this; }
        virtual InternalInterface& setHistogramBufferInfo(const DescriptorBufferInfo& histogram = {}) { histogramBufferInfo = histogram; return *this; }
        virtual InternalInterface& setPrefixScansBufferInfo(const DescriptorBufferInfo& prefixScans = {}) { prefixScansBufferInfo = prefixScans; return *this; }

        // Points the table views at the tables of the last pass over n elements.
        InternalInterface& refreshTableViews(std::size_t n) {
            const uint32_t *counts = nullptr, *offsets = nullptr, *bases = nullptr;
            uint32_t tilesPerGroup = 0, tile = 0, tiles = 0;
            if (n == 0 || radx_hip_device_tables(sorter, n, &counts, &offsets, &bases, &tilesPerGroup) != 0) {
                histogramBufferInfo = prefixScansBufferInfo = prefixGroupBasesBufferInfo = DescriptorBufferInfo();
                return *this;
            }
            check_abi(radx_hip_geometry(sorter, n, &tile, &tiles));
            const std::size_t bytes = static_cast<std::size_t>(tiles) * RADX_HIP_RADICES * sizeof(uint32_t);
            histogramBufferInfo = DescriptorBufferInfo(const_cast<uint32_t*>(counts), 0, bytes);
            prefixScansBufferInfo = DescriptorBufferInfo(const_cast<uint32_t*>(offsets), 0, bytes);
            prefixTilesPerGroup = tilesPerGroup;
            const std::size_t groups = (tiles + tilesPerGroup - 1) / tilesPerGroup;
            prefixGroupBasesBufferInfo = bases ? DescriptorBufferInfo(const_cast<uint32_t*>(bases), 0, groups * RADX_HIP_RADICES * sizeof(uint32_t))
                                               : DescriptorBufferInfo();
            return *this;
        }

        virtual InternalInterface& setMaxElementCount(const std::size_t& elementCount = 0) { maxElementCount = elementCount; return *this; }
        // reference: buildMemory(memorySize) allocates the one scratch buffer (radx_internal.hpp:40)
        virtual InternalInterface& buildMemory(int keyBytes, int valueBytes) {
            if (sorter) { check_abi(radx_hip_destroy(sorter)); sorter = nullptr; }
            this->keyBytes = keyBytes; this->valueBytes = valueBytes;
            check_abi(radx_hip_create(device->ordinal(), maxElementCount, keyBytes, valueBytes, &sorter));
            return *this;
        }
        // No descriptor sets in HIP; kept so reference call sequences compile.
        virtual InternalInterface& buildDescriptorSet() { return *this; }

        std::size_t getMaxElementCount() const { return maxElementCount; }
        std::size_t getMemorySize() const { return radx_hip_scratch_bytes(maxElementCount, keyBytes); }
        radx_hip_sorter* handle() const { return sorter; }
        int getKeyBytes() const { return keyBytes; }
        int getValueBytes() const { return valueBytes; }
    };

    // Non-owning views of the caller's buffers (reference: radx_internal.hpp:49-73).
    class InputInterface {
    protected:
        std::shared_ptr<Device> device{};
        bool built = false;

    public:
        friend class Algorithm;
        InputInterface() {}
        InputInterface(const std::shared_ptr<Device>& device) : device(device) {}
        virtual ~InputInterface() {}

        DescriptorBufferInfo keysBufferInfo{}, swapBufferInfo{};
        // Payload that travels with its key: the binding the reference declares and never wires
        // (shaders-sdk/radix/includes.glsl:151; radx_implement.inl:283,301).
        DescriptorBufferInfo valuesBufferInfo{}, valuesSwapBufferInfo{};
        std::size_t elementCount = 0;

        virtual InputInterface& setKeysBufferInfo(const DescriptorBufferInfo& keys = {}) { keysBufferInfo = keys; built = false; return *this; }
        virtual InputInterface& setSwapBufferInfo(const DescriptorBufferInfo& swap = {}) { swapBufferInfo = swap; built = false; return *this; }
        virtual InputInterface& setValuesBufferInfo(const DescriptorBufferInfo& vals = {}) { valuesBufferInfo = vals; built = false; return *this; }
        virtual InputInterface& setValuesSwapBufferInfo(const DescriptorBufferInfo& swap = {}) { valuesSwapBufferInfo = swap; built = false; return *this; }
        virtual InputInterface& setElementCount(const std::size_t& elementCount = 0) { this->elementCount = elementCount; built = false; return *this; }

        // reference: radx_implement.inl:357-388 creates buffer views and writes the descriptor set;
        // here it validates what the kernels assume about the views.
        virtual InputInterface& buildDescriptorSet() {
            auto aligned = [](const DescriptorBufferInfo& b) { return (reinterpret_cast<std::uintptr_t>(b.address()) & 15u) == 0; };
            if (elementCount > 0) {
                if (!keysBufferInfo.buffer || !swapBufferInfo.buffer)
                    throw Error(static_cast<int>(hipErrorInvalidValue), "InputInterface: keys/swap buffer not set");
                if (!aligned(keysBufferInfo) || !aligned(swapBufferInfo))
                    throw Error(static_cast<int>(hipErrorInvalidValue), "InputInterface: keys/swap must be 16-byte aligned");
                if ((valuesBufferInfo.buffer != nullptr) != (valuesSwapBufferInfo.buffer != nullptr))
                    throw Error(static_cast<int>(hipErrorInvalidValue), "InputInterface: values and values-swap go together");
                if (valuesBufferInfo.buffer && (!aligned(valuesBufferInfo) || !aligned(valuesSwapBufferInfo)))
                    throw Error(static_cast<int>(hipErrorInvalidValue), "InputInterface: value buffers must be 16-byte aligned");
            }
            built = true;
            return *this;
        }
        bool isBuilt() const { return built; }
    };

    // reference: radx_internal.hpp:77-101
    class Algorithm : public std::enable_shared_from_this<Algorithm> {
    protected:
        std::shared_ptr<Device> device{};
        virtual Result command(const CommandBuffer&, const std::unique_ptr<InternalInterface>&,
                               const std::shared_ptr<InputInterface>&, Result& res) { return res = SUCCESS; }
        virtual Result createInternalMemory(std::unique_ptr<InternalInterface>&, const std::size_t& = 1024 * 1024) { return SUCCESS; }

    public:
        Algorithm() {}
        virtual ~Algorithm() {}
        template <class T> friend class Sort;
        virtual Result initialize(const std::shared_ptr<Device>& device) { this->device = device; return SUCCESS; }
    };

    // reference: radx_internal.hpp:104-134
    template <class T>
    class Sort : public std::enable_shared_from_this<Sort<T>> {
    protected:
        std::shared_ptr<T> algorithm{};
        std::shared_ptr<Device> device{};
        std::unique_ptr<InternalInterface> internalInterface{};

    public:
        virtual ~Sort() {}

        virtual Sort<T>& initialize(const std::shared_ptr<Device>& device, const std::shared_ptr<T>& algorithm,
                                    const std::size_t& maxElementCount = 1024 * 1024) {
            this->device = device; this->algorithm = algorithm;
            this->internalInterface = std::make_unique<InternalInterface>(this->device);
            this->algorithm->createInternalMemory(this->internalInterface, maxElementCount);
            return *this;
        }

        // accepts only right-based links (reference: radx_internal.hpp:122-125)
        virtual Sort<T>& initialize(const std::shared_ptr<Device>& device, std::shared_ptr<Algorithm>&& algorithm,
                                    const std::size_t& maxElementCount = 1024 * 1024) {
            std::shared_ptr<T> typed = std::dynamic_pointer_cast<T>(std::move(algorithm));
            if (!typed) throw Error(static_cast<int>(hipErrorInvalidValue), "Sort::initialize: the algorithm is not of this Sort's type");
            return this->initialize(device, typed, maxElementCount);
        }

        // Enqueue the sort on cmdBuf's stream.  One sort in flight per Sort object.
        virtual Sort<T>& command(const CommandBuffer& cmdBuf, std::shared_ptr<InputInterface>& inputInterface) {
            Result res = SUCCESS;
            algorithm->command(cmdBuf, internalInterface, inputInterface, res);
            return *this;
        }

        const InternalInterface& getInternalInterface() const { return *internalInterface; }
        InternalInterface& getInternalInterface() { return *internalInterface; }
    };

    // reference: radx_internal.hpp:137-148, radx_implement.inl:391-490
    // reference: shaders-sdk/radix/includes.glsl:137 (declared, never read by the shaders)
    struct RadicePropStruct { uint32_t Descending = 0, IsSigned = 0; };

    class Radix : public Algorithm {
    protected:
        int keyBytes = 4, valueBytes = 0;
        RadicePropStruct radiceProp{};
        bool floatKeys = false;

        // reference: radx_implement.inl:421-447 — the pass loop, one C-ABI call here
        Result command(const CommandBuffer& cmdBuf, const std::unique_ptr<InternalInterface>& internalInterface,
                       const std::shared_ptr<InputInterface>& inputInterface, Result& res) override {
            if (!inputInterface->isBuilt()) inputInterface->buildDescriptorSet();
            const std::size_t n = inputInterface->elementCount;
            const std::size_t need = n * static_cast<std::size_t>(keyBytes);
            if (n && (inputInterface->keysBufferInfo.range < need || inputInterface->swapBufferInfo.range < need))
                throw Error(static_cast<int>(hipErrorInvalidValue), "Radix::command: keys/swap range smaller than elementCount");
            const std::size_t needValues = n * static_cast<std::size_t>(valueBytes);
            if (n && valueBytes && (inputInterface->valuesBufferInfo.range < needValues ||
                                    inputInterface->valuesSwapBufferInfo.range < needValues))
                throw Error(static_cast<int>(hipErrorInvalidValue), "Radix::command: values/valuesSwap range smaller than elementCount");
            res = radx_hip_sort(internalInterface->handle(), inputInterface->keysBufferInfo.address(),
                                inputInterface->swapBufferInfo.address(), inputInterface->valuesBufferInfo.address(),
                                inputInterface->valuesSwapBufferInfo.address(), n, cmdBuf.stream);
            check_abi(res);
            internalInterface->refreshTableViews(n);
            return res;
        }

        // reference: radx_implement.inl:450-490
        Result createInternalMemory(std::unique_ptr<InternalInterface>& internalInterface,
                                    const std::size_t& maxElementCount = 1024 * 1024) override {
            internalInterface->setMaxElementCount(maxElementCount);
            internalInterface->buildMemory(keyBytes, valueBytes).buildDescriptorSet();
            check_abi(radx_hip_set_key_order(internalInterface->handle(), floatKeys ? 2 : (radiceProp.IsSigned ? 1 : 0),
                                             radiceProp.Descending ? 1 : 0));
            return SUCCESS;
        }

    public:
        Radix() {}
        template <class T> friend class Sort;

        // Key width (4 or 8 bytes) and payload width (0 or 4).  The reference fixes these at shader
        // compile time (KEYTYPE, shaders-sdk/radix/includes.glsl:128-135); set them before
        // Sort::initialize.
        Radix& setKeyBytes(int bytes) { keyBytes = bytes; return *this; }
        Radix& setValueBytes(int bytes) { valueBytes = bytes; return *this; }
        // Key order: descending and/or signed integers (RadicePropStruct), or IEEE floats of the key
        // width.  Ties keep input order in every case.  Set before Sort::initialize.
        Radix& setRadiceProp(const RadicePropStruct& prop) { radiceProp = prop; return *this; }
        Radix& setFloatKeys(bool on) { floatKeys = on; return *this; }

        // The index path the reference builds as pipelines 3 and 4 and never dispatches
        // (radx_implement.inl:413-414; shaders-sdk/radix/indiction.comp:18-29, permutation.comp:18-29):
        //   indiction:   ptrs[i] = i; cache[i] = in[i]      permutation: in[i] = cache[ptrs[i]]
        // Sort (key, ptrs) pairs with a value_bytes = 4 sorter in between and wide payload rows
        // move exactly once.  Rows are elemBytes wide.
        Radix& indiction(const CommandBuffer& cmdBuf, const DescriptorBufferInfo& ptrs, const DescriptorBufferInfo& cache,
                         const DescriptorBufferInfo& in, std::size_t count, std::size_t elemBytes) {
            check_abi(radx_hip_indiction(static_cast<uint32_t*>(ptrs.address()), cache.address(), in.address(), count,
                                         elemBytes, cmdBuf.stream));
            return *this;
        }
        Radix& permutation(const CommandBuffer& cmdBuf, const DescriptorBufferInfo& in, const DescriptorBufferInfo& cache,
                           const DescriptorBufferInfo& ptrs, std::size_t count, std::size_t elemBytes) {
            check_abi(radx_hip_permutation(in.address(), cache.address(), static_cast<const uint32_t*>(ptrs.address()),
                                           count, elemBytes, cmdBuf.stream));
            return *this;
        }

        // reference: radx_implement.inl:391-418 builds five pipelines; the kernels here are
        // already part of the loaded code object.
        Result initialize(const std::shared_ptr<Device>& device) override { this->device = device; return SUCCESS; }
    };
}
