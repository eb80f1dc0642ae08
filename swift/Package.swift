// swift-tools-version:6.2
// HeAmd -- the Swift side of the MI355X-native BFV engine (libhe_amd.so), as a package that sits NEXT TO an unmodified
// checkout of apple/swift-homomorphic-encryption:
//
//     export HE_AMD_LIB_DIR=/opt/he_amd/lib            # where libhe_amd.so was installed (build.py output)
//     swift build -c release                            # in this directory
//
// It follows the pattern of the reference's own C target CUtil (reference Package.swift:100-116, used from
// Sources/HomomorphicEncryption/Zeroization.swift:28-39): a C target that only carries a header, and Swift code that
// imports it.  NOT compiled in the build image of this repository (it has no Swift toolchain): written against the
// reference's public API, file:line cited next to every use.
import Foundation
import PackageDescription

let libDir = ProcessInfo.processInfo.environment["HE_AMD_LIB_DIR"] ?? "/opt/he_amd/lib"
// the reference checkout: a sibling directory by default, or REFERENCE_PATH
let referencePath = ProcessInfo.processInfo.environment["SWIFT_HE_PATH"] ?? "../../swift-homomorphic-encryption"

let package = Package(
    name: "swift-homomorphic-encryption-amd",
    platforms: [.macOS(.v26)],
    products: [
        .library(name: "HeAmd", targets: ["HeAmd"]),
    ],
    dependencies: [
        .package(path: referencePath),
    ],
    targets: [
        .target(
            name: "CHeAmd",
            path: "Sources/CHeAmd",
            sources: ["shim.c"],
            publicHeadersPath: "include",
            linkerSettings: [
                .unsafeFlags(["-L\(libDir)", "-Xlinker", "-rpath", "-Xlinker", libDir]),
                .linkedLibrary("he_amd"), // links libamdhip64 itself: Swift never sees HIP
            ]),
        .target(
            name: "HeAmd",
            dependencies: [
                "CHeAmd",
                .product(name: "HomomorphicEncryption", package: "swift-homomorphic-encryption"),
                .product(name: "PrivateInformationRetrieval", package: "swift-homomorphic-encryption"),
            ],
            path: "Sources/HeAmd",
            swiftSettings: [.unsafeFlags(["-cross-module-optimization"])]), // as the reference's librarySettings
        .testTarget(
            name: "HeAmdTests",
            dependencies: [
                "HeAmd",
                .product(name: "HomomorphicEncryption", package: "swift-homomorphic-encryption"),
                .product(name: "PrivateInformationRetrieval", package: "swift-homomorphic-encryption"),
                // the reference's generic scheme / index-PIR suites (reference Package.swift:74,157), instantiated with
                // GpuBfv and GpuPirUtil in Tests/HeAmdTests/ReferenceSuites.swift
                .product(name: "_TestUtilities", package: "swift-homomorphic-encryption"),
            ],
            path: "Tests/HeAmdTests"),
    ])
